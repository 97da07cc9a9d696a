#!/bin/bash
# Round-end evidence on the GPU box, in one call: the full GPU test suite, smoke, the bench line of
# every config and the rocprofv3 passes of configs 2 and 3.  Everything lands under gpurun_out/final/;
# tools/final_copy.sh files it under profiles/.   usage: bash tools/final_run.sh <round tag, e.g. r02>
set -o pipefail
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 || { tail -20 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err && tail -1 $OUT/bench_c2.json | cut -c1-200
python bench.py --steps 20 --warmup 5 > $OUT/bench_c2_steps20.json 2>/dev/null
for c in c3 c4 c5; do
  python bench.py --config $c --steps 12 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err && tail -1 $OUT/bench_$c.json | cut -c1-160
done
python tools/skewed_queries.py > $OUT/skewed_queries.txt 2>&1 && tail -3 $OUT/skewed_queries.txt
python tools/rerank_multi_time.py > $OUT/rerank_multi.txt 2>&1 && tail -10 $OUT/rerank_multi.txt
bash tools/profile.sh ${TAG}_c2 c2 > $OUT/profile_c2.log 2>&1 && echo "profile c2 done"
bash tools/profile.sh ${TAG}_c3 c3 > $OUT/profile_c3.log 2>&1 && echo "profile c3 done"
bash tools/profile.sh ${TAG}_c5 c5 > $OUT/profile_c5.log 2>&1 && echo "profile c5 done"
