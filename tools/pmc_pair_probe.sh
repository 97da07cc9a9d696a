#!/bin/bash
# a rocprofv3 run with FETCH_SIZE and WRITE_SIZE in ONE --pmc pass hung in round 3.
# Is it the counter pair, or the product under serialised dispatch?  Step 1: the pair on a program that
# has nothing of this library in it (tools/micro/issue_rates); step 2, only if step 1 came back: the pair
# on the smallest bench configuration, kernel leg only.  Every step under its own timeout.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_pair_probe
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "step 1: FETCH_SIZE + WRITE_SIZE on tools/micro/issue_rates"
timeout -k 5 90 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/micro -o run -- $REPO/tools/micro/issue_rates > $OUT/micro.log 2>&1
rc1=$?
echo "  exit code $rc1 (124 / 137 = killed by the timeout)"; tail -2 $OUT/micro.log
ls $OUT/micro 2>/dev/null | head -3
if [ $rc1 -ne 0 ]; then echo "step 2 skipped"; exit 0; fi
echo "step 2: the pair on bench.py --config small --kernel-leg-only"
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/small -o run -- python3 $REPO/bench.py --config small --steps 4 --warmup 1 --no-cpu-baseline --check 0 --kernel-leg-only --coalesce-threads 0 > $OUT/small.log 2>&1
rc2=$?
echo "  exit code $rc2"; tail -3 $OUT/small.log
ls $OUT/small 2>/dev/null | head -3
