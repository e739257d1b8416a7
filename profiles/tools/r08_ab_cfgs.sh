cd $GRAFT_REPO_ROOT
cp era-zk_evm_amd/libzkw.so /tmp/keep.so
for R in 1 2; do for L in base fast; do
  cp era-zk_evm_amd/ab_$L.so era-zk_evm_amd/libzkw.so
  for A in "--cfg 0 --nop-only --steps 20 --warmup 5 --fuse 20 --streams 1 --commit-mask 0" "--cfg 1 --steps 20 --warmup 5 --fuse 20 --streams 1 --commit-mask 0" "--cfg 1 --steps 64 --warmup 64 --fuse 64 --streams 1 --commit-mask 0"; do
    ZKW_BENCH_NOCHECK=1 python bench.py $A --no-cpu-baseline --no-other-configs --no-host-legs --repeats 0 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L [$A] value G', round(j['value']/1e9,2), 'kernel_ms', round(j['kernel_ms'],4), 'B/cycle', round(j['roofline']['bytes_per_cycle_this_run'],1))"
  done
done; done
cp /tmp/keep.so era-zk_evm_amd/libzkw.so
