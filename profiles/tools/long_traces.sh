# cfg 2 in its second shape: 256 instances x 4096 cycles per step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P='import sys,json; [print("instances=%d cycles=%d lanes=%s fuse=%d groups=%d ms_per_step=%.4f kernel_ms=%.3f cycles/s=%.4g"%(j["config"]["instances_per_gpu"], j["config"]["cycles_per_instance"], j["config"]["lanes_per_wave"], j["config"]["batches_per_fused_launch"], j["config"]["fused_groups_in_flight"], j["ms_per_step"], j["kernel_ms"], j["value"])) for j in map(json.loads, sys.stdin)]'
for A in "--fuse 32 --steps 64 --warmup 32" "--fuse 128 --steps 256 --warmup 128" "--fuse 256 --steps 512 --warmup 256" "--fuse 64 --streams 4 --steps 512 --warmup 256"; do
  python bench.py --no-cpu-baseline --instances 256 --cycles 4096 $A 2>&1 | grep '^{\|rror' | python -c "$P"
done
python bench.py --no-cpu-baseline 2>&1 | grep '^{' | python -c "$P"
