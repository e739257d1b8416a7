# same-box A/B of kernel variants: profiles/tools/variants/libzkw_<name>.so (built locally, git-ignored) are swapped
# in for era-zk_evm_amd/libzkw.so one after the other, interleaved twice so that clock / box drift shows
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_orig.so
P='import sys,json; [print("   fuse=%d ms_per_step=%.4f kernel_ms=%.3f cycles/s=%.4g"%(j["config"]["batches_per_fused_launch"], j["ms_per_step"], j["kernel_ms"], j["value"])) for j in map(json.loads, sys.stdin)]'
for round in 1 2; do
for v in profiles/tools/variants/libzkw_*.so; do
  cp $v era-zk_evm_amd/libzkw.so; echo "$v"
  python bench.py --no-cpu-baseline 2>&1 | grep '^{' | python -c "$P"
  python bench.py --no-cpu-baseline --fuse 16 --steps 128 2>&1 | grep '^{' | python -c "$P"
done; done
cp /tmp/libzkw_orig.so era-zk_evm_amd/libzkw.so
