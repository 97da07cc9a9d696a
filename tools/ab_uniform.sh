#!/bin/bash
# A/B of the few-term kernel forms on ONE box: SLG_UNIFORM_KERNEL=3 (slg_score_uni3.hpp) vs 2
# (round-2 kernel), kernel leg only, alternating.  usage: bash tools/ab_uniform.sh [reps] [config]
REPS=${1:-2}; CFG=${2:-c2}
for rep in $(seq 1 $REPS); do for kv in 3 2; do
  out=$(SLG_UNIFORM_KERNEL=$kv python3 bench.py --config $CFG --steps 16 --warmup 2 --no-cpu-baseline --check 16 --kernel-leg-only 2>/dev/null | tail -1)
  echo "uniform_kernel=$kv $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "parity", d.get("parity"))')"
done; done
