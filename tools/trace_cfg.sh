#!/bin/bash
# Per-kernel average durations of one whole bench configuration (rocprofv3 kernel trace of bench.py).
# usage: bash tools/trace_cfg.sh <config> [steps] [label]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-c4}; STEPS=${2:-6}; LABEL=${3:-run}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/trace_cfg_${CFG}_$LABEL
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $REPO/bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --check 0 > $OUT/log 2>&1
tail -1 $OUT/log | cut -c1-200
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.5:
            print('  %-70s calls %5s avg_us %10.2f pct %s' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
