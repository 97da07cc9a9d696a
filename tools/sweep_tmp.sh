run() { echo "$1 :: $(env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --inflight 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity'))")"; }
run "SLG_X=1"
run "SLG_WIN_SHIFT=0"
run "SLG_WIN_SHIFT=2"
run "SLG_WIN_SHIFT=3"
run "SLG_WIN_SHIFT=4"
run "SLG_WIN_SHIFT=5"
