#!/bin/bash
# A/B of one environment setting on ONE box, kernel leg only, alternating:
# usage: bash tools/ab_env.sh "<VAR=value ...>" [reps] [config]   (the setting vs the defaults)
SET=$1; REPS=${2:-2}; CFG=${3:-c3}
for rep in $(seq 1 $REPS); do for t in "$SET" ""; do
  out=$(env $t python3 bench.py --config $CFG --steps 12 --warmup 2 --no-cpu-baseline --check 16 --kernel-leg-only 2>/dev/null | tail -1)
  echo "env=[$t] $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel", d["roofline"]["kernel"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "value", d["value"], "parity", d.get("parity"))')"
done; done
