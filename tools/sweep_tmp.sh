run() { echo "$1 :: $(python bench.py --steps 40 --warmup 5 --no-cpu-baseline $1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity'))")"; }
run "--inflight 1"
run "--inflight 2"
run "--inflight 3"
run "--inflight 4"
