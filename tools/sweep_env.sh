#!/bin/bash
# Sweep one environment variable over values on ONE box (kernel leg, then the whole step):
# usage: bash tools/sweep_env.sh <config> <VAR> <value> [<value> ...]     ("-" = unset)
CFG=$1; VAR=$2; shift 2
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  out=$(python3 bench.py --config $CFG --steps 16 --warmup 2 --no-cpu-baseline --check 16 2>/dev/null | tail -1)
  echo "cfg=$CFG $VAR=$v $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", round(d["value"]), "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "slices", d["config"].get("slices"), "parity", d.get("parity"))')"
done
