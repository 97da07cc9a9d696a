#!/bin/bash
# A/B of library builds on ONE box, whole bench line (value, step, kernel), alternating:
# usage: bash tools/ab_value.sh <config> <reps> <tag> [<tag> ...]   ("-" = the default build)
CFG=$1; REPS=$2; shift 2
for rep in $(seq 1 $REPS); do for t in "$@"; do
  if [ "$t" = "-" ]; then unset SLG_LIB_TAG; else export SLG_LIB_TAG=$t; fi
  out=$(python3 bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --check 16 --coalesce-threads 0 2>/dev/null | tail -1)
  echo "cfg=$CFG lib=$t $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", round(d["value"]), "ms_per_step", d["ms_per_step"], "resident", d["config"].get("kernel_only_qps"), "kernel_ms", d["roofline"]["kernel_ms"], "parity", d.get("parity"))')"
done; done
