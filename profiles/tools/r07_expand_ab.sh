# same-box A/B of era-zk_evm_amd/ab_*.so on the expand figures of the driver's command
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
for R in $(seq 1 ${2:-2}); do
for L in era-zk_evm_amd/ab_*.so; do
  cp $L era-zk_evm_amd/libzkw.so
  ZKW_BENCH_NOCHECK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --repeats 1 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); e=j['roofline']['expand']; print('$L fused_ms', round(e['fused_kernel_ms'],4), 'GBps', round(e['GBps']), 'cm_ms', round(e['cycle_major_kernel_ms'],4), 'cm_GBps', round(e['cycle_major_GBps']), 'one_ms', round(e['one_batch_kernel_ms'],4))" | tee -a $T/expand_ab.txt
done
done
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
