# PMC instruction mix / wait counters of the cycle kernel for one-stream fused launches: r02_pmc.sh <tag> [fuse]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
F=${2:-32}
CMD="python bench.py --fuse $F --streams 1 --steps $((F*3)) --warmup $F --no-cpu-baseline"
$CMD 2>/dev/null | grep '^{' > $T/bench_f$F.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM --output-format csv -d $T/pmc_insts -o x -- $CMD > $T/pmc_insts.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/pmc_wait -o x -- $CMD > $T/pmc_wait.log 2>&1
python - $T $F <<'PY'
import csv, glob, sys, os, collections, json
out, F = sys.argv[1], int(sys.argv[2])
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
wc = F * 64 * 256.0  # wave-cycles per launch
with open(os.path.join(out, "pmc.txt"), "w") as f:
    for k, v in sorted(pm.items()):
        f.write("%s dispatches %d avg %.6g per_wave_cycle %.2f\n" % (k, len(v), sum(v) / len(v), sum(v) / len(v) / wc))
print(open(os.path.join(out, "pmc.txt")).read())
j = json.loads(open(os.path.join(out, "bench_f%d.json" % F)).read())
print("kernel_ms", j["kernel_ms"], "kernel G cycles/s", j["kernel_cycles_per_s"] / 1e9)
PY
