# waves per workgroup (1 / 2 / 4) for the driver's command and the default command
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
for G in 4 2 1; do
  for CMD in "--steps 20 --warmup 5" "--steps 512 --warmup 128"; do
    ZKW_WAVES_PER_GROUP=$G python bench.py $CMD --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('wpg $G', '$CMD', 'value G', round(j['value']/1e9,2), 'kernel_ms', round(j['kernel_ms'],3), 'alone', round(j['kernel_ms_alone'],3))" | tee -a $T/wpg.txt
  done
done
