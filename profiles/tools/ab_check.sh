# quick parity + bench check used between kernel experiments (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1000 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | cut -c1-300
P='import sys,json; [print("fuse=%d groups=%d ms_per_step=%.4f kernel_ms=%.3f alone=%.3f cycles/s=%.4g"%(j["config"]["batches_per_fused_launch"],j["config"]["fused_groups_in_flight"], j["ms_per_step"], j["kernel_ms"], j["kernel_ms_alone"], j["value"])) for j in map(json.loads, sys.stdin)]'
python bench.py --no-cpu-baseline --fuse 16 --steps 128 2>&1 | grep '^{\|Error\|error' | python -c "$P"
python bench.py --no-cpu-baseline 2>&1 | grep '^{\|Error\|error' | python -c "$P"
python bench.py --no-cpu-baseline 2>&1 | grep '^{\|Error\|error' | python -c "$P"
python bench.py --no-cpu-baseline --fuse 1 --steps 32 2>&1 | grep '^{\|Error\|error' | python -c "$P"
