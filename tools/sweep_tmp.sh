for cfg in c3; do for rps in 4 8; do
echo "cfg=$cfg rps=$rps $(SLG_ROUNDS_PER_SLICE=$rps python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --check 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['slices'])")"
done; done
