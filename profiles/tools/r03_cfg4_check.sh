cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
P='import sys,json; [print("%s fuse=%d mask=%d groups=%d ms_per_step=%.4f kernel_ms=%.3f cycles/s=%.4g"%(j["config"]["workload"][:30], j["config"]["batches_per_fused_launch"], j["config"]["commit_mask"], j["config"]["fused_groups_in_flight"], j["ms_per_step"], j["kernel_ms"], j["value"])) for j in map(json.loads, sys.stdin)]'
for E in "" "ZKW_DEBUG_FLAGS=0x10000000" "ZKW_WAVES_PER_GROUP=4 ZKW_DEBUG_FLAGS=0x10000000"; do
  echo "== [$E]" | tee -a $T/cfg4.txt
  env $E python bench.py --no-cpu-baseline --cfg 4 --cycles 1024 --commit-mask 7 --fuse 16 --steps 32 --warmup 16 2>&1 | grep '^{\|rror' | python -c "$P" | tee -a $T/cfg4.txt
  env $E python bench.py --no-cpu-baseline --cfg 2 --commit-mask 7 --fuse 32 --steps 64 --warmup 32 2>&1 | grep '^{\|rror' | python -c "$P" | tee -a $T/cfg4.txt
done
