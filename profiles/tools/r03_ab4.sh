# A/B of ab_*.so on the driver / default commands + cfg 4 / cfg 2 with all commitments (chain kernel), then the GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
bash profiles/tools/r02_ab_libs.sh $1 2 > /dev/null 2>&1
cat $T/ab_libs.txt
P='import sys,json; [print("%s fuse=%d mask=%d ms_per_step=%.4f kernel_ms=%.3f cycles/s=%.4g"%(j["config"]["workload"], j["config"]["batches_per_fused_launch"], j["config"]["commit_mask"], j["ms_per_step"], j["kernel_ms"], j["value"])) for j in map(json.loads, sys.stdin)]'
cp era-zk_evm_amd/libzkw.so /tmp/keep.so
for L in era-zk_evm_amd/ab_*.so; do
  cp $L era-zk_evm_amd/libzkw.so; echo "== $L" | tee -a $T/commit7.txt
  python bench.py --no-cpu-baseline --cfg 4 --cycles 1024 --commit-mask 7 --fuse 16 --steps 32 --warmup 16 2>&1 | grep '^{\|rror' | python -c "$P" | tee -a $T/commit7.txt
  python bench.py --no-cpu-baseline --cfg 2 --commit-mask 7 --fuse 32 --steps 64 --warmup 32 2>&1 | grep '^{\|rror' | python -c "$P" | tee -a $T/commit7.txt
done
cp /tmp/keep.so era-zk_evm_amd/libzkw.so
timeout 2400 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; tail -5 $T/pytest.log
