cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r07_expand
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k expand 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --repeats 2 2>/dev/null | grep '^{' | tee gpurun_out/r07_expand/line.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value G', round(j['value']/1e9,2), 'kernel_ms', j['kernel_ms'], json.dumps(j['roofline'].get('expand')))"
