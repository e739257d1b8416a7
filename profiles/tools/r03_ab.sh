# same-box A/B of prebuilt libraries era-zk_evm_amd/ab_*.so, then the GPU parity suite on the LAST one (left as libzkw.so for the run)
# usage: r03_ab.sh <tag> [rounds] [pytest: 0/1]
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
for R in $(seq 1 ${2:-2}); do
for L in era-zk_evm_amd/ab_*.so; do
  cp $L era-zk_evm_amd/libzkw.so
  for A in "--steps 20 --warmup 5" ""; do
    python bench.py $A --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L [$A] value G', round(j['value']/1e9,2), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', round(j['kernel_ms'],3), 'frac', round(j['roofline']['frac'],3))" | tee -a $T/ab_libs.txt
  done
done
done
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
if [ "${3:-1}" = "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; tail -5 $T/pytest.log; fi
