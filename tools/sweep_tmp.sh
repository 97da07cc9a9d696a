run() { echo "T=$1 :: $(timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --terms $1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['slices'], d['parity']['bit_exact'])")"; }
for t in 6 8 16 32; do run $t; done
