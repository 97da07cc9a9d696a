#!/bin/bash
# Host-inclusive rate (bench.py `value`) for several caller-thread counts and library variants,
# repeated: the figure varies by tens of percent between runs.  usage: bash tools/host_rate.sh "tags" "threads" reps
for rep in $(seq 1 ${3:-3}); do for tag in $1; do for t in $2; do
  if [ "$tag" = "-" ]; then unset SLG_LIB_TAG; else export SLG_LIB_TAG=$tag; fi
  python bench.py --steps 128 --warmup 16 --no-cpu-baseline --check 0 --host-threads $t 2>/dev/null | python3 -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib', os.environ.get('SLG_LIB_TAG','-'), 'threads', d['config']['host_threads'], 'value', d['value'], 'ms/step', d['ms_per_step'])"
done; done; done
