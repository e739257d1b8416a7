# A/B of ab_*.so: driver / default commands (N rounds) + cfg 4 without commitments (near calls, storage), then the GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
bash profiles/tools/r02_ab_libs.sh $1 ${2:-2} > /dev/null 2>&1
cat $T/ab_libs.txt
P='import sys,json; [print("%s fuse=%d mask=%d ms_per_step=%.4f kernel_ms=%.3f cycles/s=%.4g"%(j["config"]["workload"], j["config"]["batches_per_fused_launch"], j["config"]["commit_mask"], j["ms_per_step"], j["kernel_ms"], j["value"])) for j in map(json.loads, sys.stdin)]'
cp era-zk_evm_amd/libzkw.so /tmp/keep.so
for R in 1 2; do for L in era-zk_evm_amd/ab_*.so; do
  cp $L era-zk_evm_amd/libzkw.so; echo "== $L" | tee -a $T/cfg4.txt
  python bench.py --no-cpu-baseline --cfg 4 --cycles 1024 --commit-mask 0 --fuse 32 --steps 64 --warmup 32 2>&1 | grep '^{\|rror' | python -c "$P" | tee -a $T/cfg4.txt
done; done
cp /tmp/keep.so era-zk_evm_amd/libzkw.so
timeout 2400 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; tail -4 $T/pytest.log | head -3
