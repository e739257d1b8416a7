#!/bin/bash
# Collects the measurements committed under profiles/ (run on the GPU box through gpurun):
#   profiles/collect.sh <tag>      e.g.  r01c
# 0. occupancy probe (profiles/tools/occupancy_probe.hip)           -> occupancy_probe.txt
# 1. bench.py default run (JSON line)                      -> gpurun_out/prof_<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command  -> kernel_stats CSV (avg launch duration)
# 3. PMC passes, each in its own run (gpurun refuses --pmc mixed with tracing domains):
#    instruction mix / wave cycles, FETCH_SIZE, WRITE_SIZE
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 profiles/tools/occupancy_probe.hip -o /tmp/occ_probe 2>/dev/null && /tmp/occ_probe > $OUT/occupancy_probe.txt 2>&1
python bench.py > $OUT/bench.log 2>&1; grep '^{' $OUT/bench.log > $OUT/bench.json
CMD="python bench.py --no-cpu-baseline"   # the default command itself (64 batches per fused launch, two fused groups in flight: commitments + restore on a side stream) minus the CPU leg
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_insts -o $TAG -- $CMD > $OUT/pmc_insts.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wait -o $TAG -- $CMD > $OUT/pmc_wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $CMD > $OUT/pmc_write.log 2>&1
# large-batch point of the instance sweep (where the HBM regime begins)
# single-launch scaling: one batch per launch, one stream (where the HBM regime begins as the batch grows)
for N in 256 1024 4096 16384 65536 262144; do S=10; [ $N -ge 65536 ] && S=3; python bench.py --instances $N --steps $S --warmup 1 --fuse 1 --streams 1 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/instance_sweep.jsonl; done
# fused-launch sweep on the default 4096 x 256 batch
# and the second shape of cfg 2 (256 instances x 4096 cycles; 256 batches = 1024 waves fill the chip)
for FS in "1 1" "4 1" "8 1" "16 1" "16 2" "32 1" "32 2" "48 2" "64 1" "64 2" "64 3" "96 2" "128 1" "128 2" "256 1"; do set -- $FS; python bench.py --steps 1024 --warmup 2 --fuse $1 --streams $2 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/fuse_sweep.jsonl; done
for A in "--fuse 32 --steps 64 --warmup 32" "--fuse 128 --steps 256 --warmup 128" "--fuse 256 --steps 512 --warmup 256"; do python bench.py --no-cpu-baseline --instances 256 --cycles 4096 $A 2>/dev/null | grep '^{' >> $OUT/long_traces.jsonl; done
find $OUT -name "*.csv" | head -40
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, json, collections, os
out, tag = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_" in r.get("Name", ""):
            res.setdefault("kernel_stats", {})[r["Name"].split("(")[0]] = r
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
res["pmc_avg_per_dispatch"] = {k: sum(v) / len(v) for k, v in pm.items()}
res["pmc_dispatches"] = {k: len(v) for k, v in pm.items()}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
