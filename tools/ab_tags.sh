#!/bin/bash
# A/B/C of several builds of the library on ONE box, kernel leg only, alternating:
# usage: bash tools/ab_tags.sh <config> <reps> <tag> [<tag> ...]   ("-" = the default build)
CFG=$1; REPS=$2; shift 2
for rep in $(seq 1 $REPS); do for t in "$@"; do
  if [ "$t" = "-" ]; then unset SLG_LIB_TAG; else export SLG_LIB_TAG=$t; fi
  out=$(python3 bench.py --config $CFG --steps 16 --warmup 2 --no-cpu-baseline --check 16 --kernel-leg-only 2>/dev/null | tail -1)
  echo "cfg=$CFG lib=$t $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "parity", d.get("parity"))')"
done; done
