# old (r01, one wave per SIMD) vs new kernel on a NOP tape, cfg1 and cfg2; PMC instruction mix of the new kernel
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/r02d; mkdir -p $T
cp era-zk_evm_amd/libzkw.so /tmp/new.so
for LIB in new r01; do
  [ $LIB = r01 ] && cp era-zk_evm_amd/libzkw_r01.so era-zk_evm_amd/libzkw.so
  for C in "--cfg 0 --nop-only" "--cfg 1 --commit-mask 0" "--cfg 2"; do
    for F in 16 32; do
      python bench.py $C --fuse $F --streams 1 --steps $((F*4)) --warmup $F --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$LIB', '$C', $F, 'kernel_ms', j['kernel_ms'], 'kcps', j['kernel_cycles_per_s']/1e9, 'value', j['value']/1e9)" >> $T/compare.txt
    done
  done
done
cp /tmp/new.so era-zk_evm_amd/libzkw.so
CMD="python bench.py --fuse 16 --streams 1 --steps 64 --warmup 16 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM --output-format csv -d $T/pmc_insts -o x -- $CMD > $T/pmc_insts.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/pmc_wait -o x -- $CMD > $T/pmc_wait.log 2>&1
python - $T <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc.txt"), "w") as f:
    for k, v in sorted(pm.items()):
        f.write("%s %d %.6g\n" % (k, len(v), sum(v) / len(v)))
print(open(os.path.join(out, "pmc.txt")).read())
PY
cat $T/compare.txt
