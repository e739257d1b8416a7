# what the witness stores cost a wave, and as what: wave cycles / wait / issue counters (PMC) of the driver's command with
# the CycleRecord stores (flag 1), the query-stream stores (2) or both (3) left out (ablations: the results are WRONG)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
for F in 0 1 2 3; do
  ZKW_DEBUG_FLAGS=$F rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $T/pmc_$F -o x -- $CMD > $T/pmc_$F.log 2>&1
  grep '^{' $T/pmc_$F.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('flags $F: value G', round(j['value']/1e9,2), 'kernel_ms', round(j['kernel_ms'],3))" | tee -a $T/ablation.txt
  python - $T/pmc_$F <<'PY' | tee -a $T/ablation.txt
import csv, glob, sys, os, collections
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
wc = 20 * 64 * 256.0
print("   " + "  ".join("%s %.0f" % (k.replace("SQ_", ""), sum(v[len(v)//2:]) / len(v[len(v)//2:]) / wc) for k, v in sorted(pm.items())))
PY
done
