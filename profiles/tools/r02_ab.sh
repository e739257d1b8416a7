# same-box A/B of environment switches for the driver's command and the default command
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
run() { # label, env, args
  env $2 python bench.py $3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1', '| value G', round(j['value']/1e9,2), 'kernel_ms', round(j['kernel_ms'],3), 'alone', round(j['kernel_ms_alone'],3))" | tee -a $T/ab.txt
}
for i in 1 2 3; do
  run "driver inline" "X=1" "--steps 20 --warmup 5"
  run "driver kernels" "ZKW_NO_INLINE_DECOMMIT=1" "--steps 20 --warmup 5"
done
run "default" "X=1" ""
run "default one-stream" "X=1" "--streams 1"
run "default one-stream no-inline" "ZKW_NO_INLINE_DECOMMIT=1" "--streams 1"
