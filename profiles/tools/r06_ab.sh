# same-box A/B of prebuilt libraries era-zk_evm_amd/ab_*.so on the driver's command (and optionally the default command), alternating rounds
# usage: r06_ab.sh <tag> [rounds] [default command too: 0/1] [pytest on the last: 0/1]
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
LIBS=(era-zk_evm_amd/ab_*.so); NL=${#LIBS[@]}
for R in $(seq 1 ${2:-3}); do
# (the order rotates from round to round: a box that warms up or drifts during the run otherwise favours the libraries that come late)
for I in $(seq 0 $((NL-1))); do L=${LIBS[$(( (I + R) % NL ))]}
  cp $L era-zk_evm_amd/libzkw.so
  CMDS=("--steps 20 --warmup 5")
  [ "${3:-0}" = "1" ] && CMDS+=("")
  for A in "${CMDS[@]}"; do
    ZKW_BENCH_NOCHECK=1 python bench.py $A --no-cpu-baseline --no-other-configs --no-host-legs --repeats 2 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L [$A] value G', round(j['value_median']/1e9,2), 'kernel_ms', round(j['kernel_ms'],4), 'frac', round(j['roofline']['frac'],3))" | tee -a $T/ab_libs.txt
  done
done
done
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
python3 - $T/ab_libs.txt <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.match(r"(\S+) \[(.*?)\] value G ([\d.]+) kernel_ms ([\d.]+)", ln)
    if m: d[(m.group(1).split('/')[-1], m.group(2))].append((float(m.group(3)), float(m.group(4))))
for k, v in sorted(d.items()):
    ks = sorted(x[1] for x in v); vs = sorted(x[0] for x in v)
    print("%-28s [%s] n=%d kernel_ms median %.4f min %.4f | value median %.2f" % (k[0], k[1], len(v), ks[len(ks)//2], ks[0], vs[len(vs)//2]))
PY
if [ "${4:-0}" = "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; grep -n "passed\|failed" $T/pytest.log | tail -2; fi
