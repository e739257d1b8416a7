cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r07
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r07/pytest_gpu.log 2>&1
grep -n "passed\|failed" gpurun_out/r07/pytest_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for A in "--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs" "--no-cpu-baseline --no-other-configs"; do python bench.py $A 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value G', round(j['value']/1e9,2), round(j['value_median']/1e9,2), 'kernel_ms', round(j['kernel_ms'],4), 'frac', round(j['roofline']['frac'],3))"; done
