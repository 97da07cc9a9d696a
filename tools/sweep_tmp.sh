run() { echo "$1 :: $(env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --check 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['slices'])")"; }
run "SLG_SLICES_PER_SUBQUERY=6"
run "SLG_SLICES_PER_SUBQUERY=8"
run "SLG_SLICES_PER_SUBQUERY=12"
run "SLG_SLICES_PER_SUBQUERY=16"
run "SLG_SLICES_PER_SUBQUERY=24"
