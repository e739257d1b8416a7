cd $GRAFT_REPO_ROOT
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
for R in 1 2; do for L in era-zk_evm_amd/ab_*.so; do cp $L era-zk_evm_amd/libzkw.so; ZKW_BENCH_NO_OTHER_CONFIGS=1 python bench.py --cfg 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L', round(j['value']/1e9,1), 'GB/s kernel_ms', round(j['kernel_ms'],3), j['roofline']['lone_batch_kernel_ms'], round(j['roofline']['keccak_f_per_s']/1e9,2))"; done; done
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
timeout 900 python -m pytest tests -m gpu -x -q -k "keccak or cfg3 or precompile or ecrecover" 2>&1 | tail -2
