#!/bin/bash
# Files what tools/final_run.sh produced (merged back into gpurun_out/) under profiles/.
# usage: bash tools/final_copy.sh <round tag>
TAG=${1:-r04}
G=gpurun_out
for c in c2 c3 c4 c5 mf mf_many_term_kernel; do [ -s $G/final/bench_$c.json ] && tail -1 $G/final/bench_$c.json > profiles/${TAG}_bench_$c.json; done
[ -s $G/final/bench_c2_steps20.json ] && tail -1 $G/final/bench_c2_steps20.json > profiles/${TAG}_bench_c2_steps20.json
for c in c2 c3 mf c5; do
  [ -f $G/prof_${TAG}_$c/trace/run_kernel_stats.csv ] || continue
  cp $G/prof_${TAG}_$c/trace/run_kernel_stats.csv profiles/${TAG}_${c}_kernel_stats.csv
  cp $G/prof_${TAG}_$c/summary.txt profiles/${TAG}_${c}_rocprof_summary.txt
  cp $G/prof_${TAG}_$c/traffic.json profiles/traffic_$c.json
done
[ -s $G/final/pytest_gpu.log ] && tail -2 $G/final/pytest_gpu.log > profiles/${TAG}_pytest_gpu_tail.txt
[ -s $G/final/skewed_queries.txt ] && grep -v amdgpu.ids $G/final/skewed_queries.txt > profiles/${TAG}_skewed_queries.txt
[ -s $G/final/rerank_multi.txt ] && grep -v amdgpu.ids $G/final/rerank_multi.txt > profiles/${TAG}_rerank_multi.txt
[ -f $G/prof_${TAG}_c5/trace/run_kernel_stats.csv ] && cp $G/prof_${TAG}_c5/trace/run_kernel_stats.csv profiles/${TAG}_c5_kernel_stats.csv
[ -d $G/prof_${TAG}_c5 ] && python3 tools/pmc_avg.py $G/prof_${TAG}_c5 rerank_kernel > profiles/${TAG}_c5_rerank_pmc.txt
ls profiles | grep ${TAG}
