#!/bin/bash
# Round-end evidence on the GPU box, in three calls (a gpurun call is limited to 20 minutes):
#   part a: the full GPU test suite, smoke, the bench lines of configs 2 (default and the driver's 20 steps),
#           mf and 5, the skewed-query and multi-clause rerank tables
#   part b: the bench lines of configs 3 and 4 and the rocprofv3 passes (kernel trace + PMC) of configs 2 and mf
#   part c: the rocprofv3 passes of configs 3 and 5
# Everything lands under gpurun_out/final/; tools/final_copy.sh files it under profiles/.
# usage: bash tools/final_run.sh <round tag, e.g. r04> <a|b|c>
set -o pipefail
TAG=${1:-r04}; PART=${2:-a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
if [ "$PART" = "a" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 || { tail -20 $OUT/pytest_gpu.log; exit 1; }
  tail -2 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
  tail -1 $OUT/smoke.log
  python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err && tail -1 $OUT/bench_c2.json | cut -c1-200
  python bench.py --steps 20 --warmup 5 > $OUT/bench_c2_steps20.json 2>/dev/null && tail -1 $OUT/bench_c2_steps20.json | cut -c1-120
  for c in mf c5; do
    python bench.py --config $c --steps 16 --warmup 4 > $OUT/bench_$c.json 2> $OUT/bench_$c.err && tail -1 $OUT/bench_$c.json | cut -c1-160
  done
  SLG_NO_UNIFORM_PLANS=1 python bench.py --config mf --steps 16 --warmup 4 --no-cpu-baseline > $OUT/bench_mf_many_term_kernel.json 2>/dev/null
  python tools/skewed_queries.py > $OUT/skewed_queries.txt 2>&1 && tail -3 $OUT/skewed_queries.txt
  python tools/rerank_multi_time.py > $OUT/rerank_multi.txt 2>&1 && tail -10 $OUT/rerank_multi.txt
elif [ "$PART" = "b" ]; then
  for c in c3 c4; do
    python bench.py --config $c --steps 12 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err && tail -1 $OUT/bench_$c.json | cut -c1-160
  done
  bash tools/profile.sh ${TAG}_c2 c2 > $OUT/profile_c2.log 2>&1 && echo "profile c2 done"
  bash tools/profile.sh ${TAG}_mf mf > $OUT/profile_mf.log 2>&1 && echo "profile mf done"
else
  bash tools/profile.sh ${TAG}_c3 c3 > $OUT/profile_c3.log 2>&1 && echo "profile c3 done"
  bash tools/profile.sh ${TAG}_c5 c5 > $OUT/profile_c5.log 2>&1 && echo "profile c5 done"
fi
