W=$1
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $1 2>/dev/null | grep '^{' | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['value']/1e9,3), 'kernel_ms', round(j['kernel_ms'],4))"; }
run "--min-warmup-s $W"; run "--min-warmup-s 0.6"; run "--min-warmup-s 0.6"; run "--min-warmup-s $W"
