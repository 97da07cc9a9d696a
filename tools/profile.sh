#!/bin/bash
# Profile bench.py (config 2) on the GPU box: kernel trace + stats first, then PMC passes, each
# in its own run (never --pmc together with a trace domain).  Writes under gpurun_out/prof_<tag>/.
# usage: bash tools/profile.sh <tag>
set -o pipefail
TAG=${1:-r1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --check 0 --inflight 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c2 -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_a -o c2 -- python3 $ARGS > $OUT/pmc_a.log 2>&1
echo "pmc_a rc=$?" >> $OUT/pmc_a.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_b -o c2 -- python3 $ARGS > $OUT/pmc_b.log 2>&1
echo "pmc_b rc=$?" >> $OUT/pmc_b.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_c -o c2 -- python3 $ARGS > $OUT/pmc_c.log 2>&1
echo "pmc_c rc=$?" >> $OUT/pmc_c.log
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_d -o c2 -- python3 $ARGS > $OUT/pmc_d.log 2>&1
echo "pmc_d rc=$?" >> $OUT/pmc_d.log
# the default bench command (two batches in flight: kernels of different batches overlap, so the
# per-kernel durations in this trace are stretched by the co-running kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -o c2 -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --check 0 > $OUT/trace_default.log 2>&1
echo "trace_default rc=$?" >> $OUT/trace_default.log
cd $REPO && python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
