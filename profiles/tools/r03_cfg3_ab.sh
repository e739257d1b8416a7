# cfg 3 (precompile-dominant): A/B of ab_*.so — lone batch and fused launches — then the cfg-3 / KAT parity tests on libzkw.so
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
P='import sys,json
for j in map(json.loads, sys.stdin):
    r=j["roofline"]; print("fuse=%d steps=%d ms_per_step=%.3f kernel_ms=%.3f lone_launch_ms=%.3f msg GB/s %.2f frac %.4f keccak-f/s %.3g sha/s %.3g" % (j["config"]["batches_per_fused_launch"], j["steps"], j["ms_per_step"], j["kernel_ms"], r["lone_launch_ms"], j["value"]/1e9, r["frac"], r["keccak_f_per_s"], r["sha256_compressions_per_s"]))'
cp era-zk_evm_amd/libzkw.so /tmp/keep.so
for L in era-zk_evm_amd/ab_*.so; do
  cp $L era-zk_evm_amd/libzkw.so; echo "== $L" | tee -a $T/cfg3.txt
  for F in 1 16 128; do
    python bench.py --no-cpu-baseline --cfg 3 --commit-mask 0 --fuse $F --steps $((F*2)) --warmup $F --streams 1 2>&1 | grep '^{\|rror' | python -c "$P" | tee -a $T/cfg3.txt
  done
done
cp /tmp/keep.so era-zk_evm_amd/libzkw.so
timeout 1500 python -m pytest tests -m gpu -x -q -k "cfg3 or keccak or precompile or ecrecover" > $T/pytest.log 2>&1; tail -3 $T/pytest.log | head -2
python profiles/tools/precompile_campaign.py 0x3300 ${2:-24} 2>&1 | tail -3 | tee $T/campaign.txt
