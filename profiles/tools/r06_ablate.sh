# kernel-time ablations on the DRIVER's command (results are wrong by construction: bench's own checks are skipped with ZKW_BENCH_NOCHECK)
# usage: r06_ablate.sh <tag> "<flags> <flags> ..."
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
# the ablation tests exist only in a -DZKW_ABLATION build (built on the box into a scratch copy of libzkw.so)
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
python -c "
import sys; sys.path.insert(0,'.')
import era_zk_evm_amd
from era_zk_evm_amd import build as b
b.build_lib(force=True, extra_flags=['-DZKW_ABLATION'])"
trap 'cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so' EXIT
for r in 1 2; do
for F in $2; do
  ZKW_BENCH_NOCHECK=1 ZKW_DEBUG_FLAGS=$F python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --repeats 0 2>$T/err_$F.log | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('flags $F kernel_ms', round(j['kernel_ms'],4), 'ms/step', round(j['ms_per_step'],5))" | tee -a $T/ablate.txt
done
done
