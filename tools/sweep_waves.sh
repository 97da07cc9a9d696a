#!/bin/bash
# kernel time of the few-term kernel against the number of persistent waves per SIMD (SLG_SCORE_WAVES), ONE box
# usage: bash tools/sweep_waves.sh <config> <waves> [<waves> ...]
CFG=$1; shift
for wv in "$@"; do
  out=$(SLG_SCORE_WAVES=$wv python3 bench.py --config $CFG --steps 12 --warmup 2 --no-cpu-baseline --check 16 --kernel-leg-only --coalesce-threads 0 2>/dev/null | tail -1)
  echo "cfg=$CFG waves_per_simd=$wv $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "parity", d.get("parity"))')"
done
