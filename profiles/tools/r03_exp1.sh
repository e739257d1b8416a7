# Round-3 first look: issue-rate probe + instruction-cache / wait counters of the cycle kernel on the driver's command.
# usage: r03_exp1.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
hipcc --offload-arch=gfx950 -O3 profiles/tools/issue_probe.hip -o /tmp/issue_probe 2>/dev/null && /tmp/issue_probe > $T/issue_probe.txt 2>&1
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
$CMD 2>/dev/null | grep '^{' > $T/bench_driver.json
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $T/pmc_icache -o x -- $CMD > $T/pmc_icache.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES --output-format csv -d $T/pmc_active -o x -- $CMD > $T/pmc_active.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $T/pmc_insts -o x -- $CMD > $T/pmc_insts.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM --output-format csv -d $T/pmc_level -o x -- $CMD > $T/pmc_level.log 2>&1
python - $T <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
wc = 20 * 64 * 256.0
with open(os.path.join(out, "pmc.txt"), "w") as f:
    for k, v in sorted(pm.items()):
        v = v[len(v) // 2:]  # skip warm-up launches
        f.write("%s dispatches %d avg %.6g per_wave_cycle %.2f\n" % (k, len(v), sum(v) / len(v), sum(v) / len(v) / wc))
print(open(os.path.join(out, "pmc.txt")).read())
PY
cat $T/issue_probe.txt; cat $T/bench_driver.json | head -c 600
