#!/bin/bash
# Device-resident throughput (pre-planned batches) for several numbers of batches in flight, and the
# host-inclusive value for several caller-thread counts.  usage: bash tools/inflight_sweep.sh
for n in 1 2 3 4 8; do
  python3 bench.py --steps 64 --warmup 8 --inflight $n --host-threads 8 --regions 3 --no-cpu-baseline --check 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight', $n, 'resident_qps', d['config']['kernel_only_qps'], 'ms', d['config']['kernel_only_ms_per_step'], '| value', d['value'], d['value_spread'])"
done
for t in 4 12 16; do
  python3 bench.py --steps 64 --warmup 8 --host-threads $t --regions 3 --no-cpu-baseline --check 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('host_threads', $t, 'value', d['value'], d['value_spread'])"
done
