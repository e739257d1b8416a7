# kernel-time ablations on the DRIVER's command (results are wrong by construction: bench's own checks are skipped with ZKW_BENCH_NOCHECK)
# usage: r06_ablate.sh <tag> "<flags> <flags> ..."
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
for r in 1 2; do
for F in $2; do
  ZKW_BENCH_NOCHECK=1 ZKW_DEBUG_FLAGS=$F python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$T/err_$F.log | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('flags $F kernel_ms', round(j['kernel_ms'],4), 'ms/step', round(j['ms_per_step'],5))" | tee -a $T/ablate.txt
done
done
