# same-box A/B of compile-time variants of libzkw.so (built on the box into the tree, the shipped build restored at the end):
#   r02_variant.sh <tag> "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the shipped flags)
# for each variant: a parity subset, the driver's command twice, the default command
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T; shift
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
for FL in "$@"; do
  echo "== variant [$FL]" | tee -a $T/variants.txt
  python - "$FL" <<'PY' 2>&1 | tail -2
import sys; sys.path.insert(0, '.')
import era_zk_evm_amd
from era_zk_evm_amd import build as b
b.build_lib(force=True, extra_flags=sys.argv[1].split())
PY
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cfg2 or cfg4 or fuzz or queue_commit" 2>&1 | tail -2 | tee -a $T/variants.txt
  for A in "--steps 20 --warmup 5" "--steps 20 --warmup 5" ""; do
    python bench.py $A --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('  [$A] value G', round(j['value']/1e9,2), 'kernel_ms', round(j['kernel_ms'],3), 'alone', round(j['kernel_ms_alone'],3), 'frac', round(j['roofline']['frac'],3))" | tee -a $T/variants.txt
  done
done
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
