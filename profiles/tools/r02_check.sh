# usage: r02_check.sh <tag>  — GPU parity tests + the driver's bench command + the default bench command
set -x
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1
mkdir -p $T
python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; tail -5 $T/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $T/driver.json 2> $T/driver.err
python bench.py --no-cpu-baseline > $T/default.json 2>> $T/driver.err
for f in $T/driver.json $T/default.json; do python3 -c "
import json,sys
j=json.loads(open('$f').read().strip().split('\n')[-1])
print({k:j[k] for k in ['value','ms_per_step','kernel_ms','kernel_ms_alone','kernel_cycles_per_s']}, j['roofline']['frac'])
"; done
