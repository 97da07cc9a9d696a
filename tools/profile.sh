#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats first, then PMC passes, each in its own
# run (never --pmc together with a trace domain).  Every run is `bench.py --kernel-leg-only`: the
# scoring kernel of 8 rotating batches (posting working set >> the 256 MiB Infinity Cache), one
# launch at a time, so per-kernel averages are those of the kernel alone.
# Writes under gpurun_out/prof_<tag>/.   usage: bash tools/profile.sh <tag> [config]
set -o pipefail
TAG=${1:-r2}
CFG=${2:-c2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --config $CFG --steps 24 --warmup 2 --no-cpu-baseline --check 0 --kernel-leg-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_a -o run -- python3 $ARGS > $OUT/pmc_a.log 2>&1
echo "pmc_a rc=$?" >> $OUT/pmc_a.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_b -o run -- python3 $ARGS > $OUT/pmc_b.log 2>&1
echo "pmc_b rc=$?" >> $OUT/pmc_b.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_c -o run -- python3 $ARGS > $OUT/pmc_c.log 2>&1
echo "pmc_c rc=$?" >> $OUT/pmc_c.log
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_d -o run -- python3 $ARGS > $OUT/pmc_d.log 2>&1
echo "pmc_d rc=$?" >> $OUT/pmc_d.log
cd $REPO && python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
