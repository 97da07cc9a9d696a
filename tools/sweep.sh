#!/bin/bash
# Kernel-time sweep of planner knobs on the GPU box (bench.py --kernel-leg-only per setting).
# usage: bash tools/sweep.sh "VAR=a VAR=b,VAR2=c ..." [config]   (each word is one environment setting;
# commas join several variables into one setting; SLG_LIB_TAG=<tag> selects an experiment build)
CFG=${2:-c2}
for kv in $1; do
  out=$(env ${kv//,/ } python3 bench.py --config $CFG --steps 16 --warmup 2 --no-cpu-baseline --check 0 --kernel-leg-only 2>/dev/null | tail -1)
  echo "$kv kernel_ms=$(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "slices", d["config"]["slices"], "resident_qps", d["config"]["kernel_only_qps"])')"
done
