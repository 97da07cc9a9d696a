#!/bin/bash
# rocprofv3 kernel stats of one bench config (kernel leg only): usage: bash tools/trace_only.sh <config> [lib tag]
CFG=$1; TAG=${2:--}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
if [ "$TAG" != "-" ]; then export SLG_LIB_TAG=$TAG; fi
rm -rf $R/gpurun_out/trace_${CFG}_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_${CFG}_$TAG -o run -- python3 $R/bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline --check 0 --kernel-leg-only --coalesce-threads 0 > /dev/null 2>&1
echo "== $CFG lib=$TAG"; grep -E "merge_topk|score_uniform4|select_topk|partition" $R/gpurun_out/trace_${CFG}_$TAG/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
