# PMC (instruction mix, wave cycles, waits) of the driver's command under kernel ablation flags (results WRONG by construction)
# usage: r06_pmc_ablate.sh <tag> "<flags> ..."
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
# the ablation tests exist only in a -DZKW_ABLATION build (built on the box into a scratch copy of libzkw.so)
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
python -c "
import sys; sys.path.insert(0,'.')
import era_zk_evm_amd
from era_zk_evm_amd import build as b
b.build_lib(force=True, extra_flags=['-DZKW_ABLATION'])"
trap 'cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so' EXIT
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --repeats 0"
for F in $2; do
  for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
    N=$(echo $P | md5sum | cut -c1-6)
    ZKW_BENCH_NOCHECK=1 ZKW_DEBUG_FLAGS=$F rocprofv3 --pmc $P --output-format csv -d $T/pmc_${F}_$N -o x -- $CMD > $T/pmc_${F}_$N.log 2>&1
  done
  python - $T $F <<'PY' | tee -a $T/pmc_ablation.txt
import csv, glob, sys, os, collections
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_%s_*" % sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
wc = 20 * 64 * 256.0
print("flags %s: " % sys.argv[2] + "  ".join("%s %.1f" % (k.replace("SQ_", ""), sum(v[len(v)//2:]) / len(v[len(v)//2:]) / wc) for k, v in sorted(pm.items())))
PY
done
