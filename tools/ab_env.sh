#!/bin/bash
# A/B of one environment switch on ONE box, kernel leg only, alternating:
# usage: bash tools/ab_env.sh <VAR> <reps> <config> [<config> ...]    (VAR unset vs VAR=1)
VAR=$1; REPS=$2; shift 2
for cfg in "$@"; do for rep in $(seq 1 $REPS); do for v in 0 1; do
  if [ "$v" = "0" ]; then unset $VAR; else export $VAR=1; fi
  out=$(python3 bench.py --config $cfg --steps 16 --warmup 2 --no-cpu-baseline --check 16 --kernel-leg-only 2>/dev/null | tail -1)
  echo "cfg=$cfg $VAR=$v $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "parity", d.get("parity"))')"
done; done; done
