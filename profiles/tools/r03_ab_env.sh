# same-box A/B of one library under different environment switches, then (optionally) the GPU parity suite
# usage: r03_ab_env.sh <tag> <rounds> <pytest 0/1> "<ENV_A>" "<ENV_B>" ...     (an empty string = no switch)
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T; R=$2; PT=$3; shift 3
for r in $(seq 1 $R); do
for E in "$@"; do
  for A in "--steps 20 --warmup 5" ""; do
    env $E python bench.py $A --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('[$E] [$A] value G', round(j['value']/1e9,2), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', round(j['kernel_ms'],3), 'frac', round(j['roofline']['frac'],3))" | tee -a $T/ab_env.txt
  done
done
done
if [ "$PT" = "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; grep -n "passed\|failed\|error" $T/pytest.log | tail -5; fi
