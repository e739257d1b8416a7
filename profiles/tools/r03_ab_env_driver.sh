# same-box A/B of environment switches on the DRIVER's command only, many alternating rounds
# usage: r03_ab_env_driver.sh <tag> <rounds> "<ENV_A>" "<ENV_B>" ...
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T; R=$2; shift 2
for r in $(seq 1 $R); do
for E in "$@"; do
  env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('[$E] value G', round(j['value']/1e9,2), 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', round(j['kernel_ms'],3), 'alone', round(j['kernel_ms_alone'],3))" | tee -a $T/ab_env.txt
done
done
python3 - $T/ab_env.txt <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.match(r"\[(.*?)\] value G ([\d.]+) ms/step ([\d.]+) kernel_ms ([\d.]+)", ln)
    if m: d[m.group(1)].append((float(m.group(2)), float(m.group(4))))
for k, v in d.items():
    vs = sorted(x[0] for x in v); ks = sorted(x[1] for x in v)
    print("[%s] n=%d value median %.2f mean %.2f | kernel_ms median %.3f" % (k, len(v), vs[len(vs)//2], sum(vs)/len(vs), ks[len(ks)//2]))
PY
