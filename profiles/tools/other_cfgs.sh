# the parity configurations on the final kernel (not bench lines): cfg 1, cfg 4 (with / without all queue commitments), cfg 3
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P='import sys,json; [print("%s fuse=%d mask=%d ms_per_step=%.4f kernel_ms=%.3f cycles/s=%.4g"%(j["config"]["workload"], j["config"]["batches_per_fused_launch"], j["config"]["commit_mask"], j["ms_per_step"], j["kernel_ms"], j["value"])) for j in map(json.loads, sys.stdin)]'
python bench.py --no-cpu-baseline --cfg 1 --commit-mask 0 2>&1 | grep '^{\|rror' | python -c "$P"
python bench.py --no-cpu-baseline --cfg 4 --cycles 1024 --commit-mask 0 --fuse 32 --steps 64 --warmup 32 2>&1 | grep '^{\|rror' | python -c "$P"
python bench.py --no-cpu-baseline --cfg 4 --cycles 1024 --commit-mask 7 --fuse 16 --steps 32 --warmup 16 2>&1 | grep '^{\|rror' | python -c "$P"
python bench.py --no-cpu-baseline --cfg 2 --commit-mask 7 --fuse 32 --steps 64 --warmup 32 2>&1 | grep '^{\|rror' | python -c "$P"
python bench.py --no-cpu-baseline --cfg 3 --cycles 64 --commit-mask 0 --fuse 16 --steps 32 --warmup 16 2>&1 | grep '^{\|rror' | python -c "$P"
