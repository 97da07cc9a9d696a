#!/bin/bash
# Per-kernel average durations (rocprofv3 kernel trace of `bench.py --kernel-leg-only`) for several
# library variants on one box.  usage: bash tools/trace_ab.sh "tags" [config]   ("-" = the default library)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${2:-c2}
cd /tmp && export TMPDIR=/tmp
for tag in $1; do
  if [ "$tag" = "-" ]; then unset SLG_LIB_TAG; else export SLG_LIB_TAG=$tag; fi
  OUT=$REPO/gpurun_out/trace_ab_$tag
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $REPO/bench.py --config $CFG --steps 24 --warmup 2 --no-cpu-baseline --check 0 --kernel-leg-only > $OUT/log 2>&1
  echo "== $tag"
  python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r['Name'] for s in ('score_', 'partition', 'merge', 'select')):
            print('  %-60s calls %5s avg_us %9.2f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
