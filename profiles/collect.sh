#!/bin/bash
# Collects the measurements committed under profiles/ (run on the GPU box through gpurun):
#   profiles/collect.sh <tag>      e.g.  r02a
# 0. occupancy probe (profiles/tools/occupancy_probe.hip)                     -> occupancy_probe.txt
# 1. bench.py, the DRIVER's command (--steps 20 --warmup 5) and the default   -> bench_driver.json, bench.json
# 2. rocprofv3 --kernel-trace --stats of both commands                        -> kernel_stats CSVs (avg launch duration)
# 3. PMC passes on the driver's command, each in its own run (gpurun refuses --pmc mixed with tracing domains):
#    instruction mix / wave cycles / waits; HBM traffic + ablations: profiles/tools/r02_traffic.sh
# 4. sweeps: instances per launch, batches per fused launch, the second shape of cfg 2, the other configurations
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export ZKW_BENCH_NO_OTHER_CONFIGS=1   # every bench.py below measures ITS workload only; the driver's full line is taken with the switch unset
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -Wno-unused-result profiles/tools/occupancy_probe.hip -o /tmp/occ_probe 2>/dev/null && /tmp/occ_probe > $OUT/occupancy_probe.txt 2>&1
DRIVER="python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-host-legs"   # (the traced / counted process runs the headline workload only: its rocprof averages then describe one launch shape)
# The published bench line of each command is the one printed by the SAME process rocprofv3 traced (kernel-trace only:
# its overhead is not measurable here), so that the HIP-event duration in the line and the rocprof average describe the
# same launches: processes on one box differ by up to 6 % from each other (15.05 vs 15.98 G in the r03c collection,
# same library, seconds apart), boxes by +-3 %.  The untraced runs before them are kept as *_plain.json.
# the driver's command exactly as the driver runs it (cpu_baseline, other_configs, 5 timed regions): the line BENCH_rNN.json will carry
env -u ZKW_BENCH_NO_OTHER_CONFIGS python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_full.log 2>&1; grep '^{' $OUT/bench_driver_full.log > $OUT/bench_driver_full.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-host-legs > $OUT/bench_driver_plain.log 2>&1; grep '^{' $OUT/bench_driver_plain.log > $OUT/bench_driver_plain.json
python bench.py --no-cpu-baseline --no-other-configs --no-host-legs > $OUT/bench_plain.log 2>&1; grep '^{' $OUT/bench_plain.log > $OUT/bench_plain.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -o $TAG -- $DRIVER > $OUT/trace_driver.log 2>&1; grep '^{' $OUT/trace_driver.log > $OUT/bench_driver.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python bench.py --no-other-configs --no-host-legs > $OUT/trace.log 2>&1; grep '^{' $OUT/trace.log > $OUT/bench.json
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_insts -o $TAG -- $DRIVER --no-cpu-baseline > $OUT/pmc_insts.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wait -o $TAG -- $DRIVER --no-cpu-baseline > $OUT/pmc_wait.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $DRIVER --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $DRIVER --no-cpu-baseline > $OUT/pmc_write.log 2>&1
# the other single-GPU BASELINE configurations, each alone in a traced process with the arguments bench.py's other_configs uses
# (bench.py: OTHER_CONFIGS): one kernel-stats CSV and one line per configuration
i=0
for A in "--cfg 1 --instances 256 --cycles 256 --steps 20 --warmup 20 --fuse 20 --streams 1 --commit-mask 0" \
         "--cfg 1 --instances 4096 --cycles 256 --steps 64 --warmup 64 --fuse 64 --streams 1 --commit-mask 0" \
         "--cfg 3 --instances 512 --steps 256 --warmup 256 --fuse 256 --streams 1 --commit-mask 0" \
         "--cfg 4 --instances 4096 --cycles 1024 --steps 32 --warmup 16 --fuse 16 --streams 2 --commit-mask 7"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_oc$i -o $TAG -- python bench.py $A --no-cpu-baseline --min-warmup-s 0.2 > $OUT/trace_oc$i.log 2>&1
  grep '^{' $OUT/trace_oc$i.log >> $OUT/other_configs_traced.jsonl
  cp $OUT/trace_oc$i/${TAG}_kernel_stats.csv $OUT/oc${i}_kernel_stats.csv 2>/dev/null || find $OUT/trace_oc$i -name "*kernel_stats.csv" -exec cp {} $OUT/oc${i}_kernel_stats.csv \;
  i=$((i+1))
done
# single-launch scaling: one batch per launch, one stream
for N in 256 1024 4096 16384 65536 131072 262144; do S=10; [ $N -ge 65536 ] && S=3; python bench.py --instances $N --steps $S --warmup 1 --fuse 1 --streams 1 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/instance_sweep.jsonl; done
# fused-launch sweep on the default 4096 x 256 batch (batches per launch, groups in flight)
for FS in "1 1" "4 1" "8 1" "16 1" "20 1" "32 1" "32 2" "48 2" "64 1" "64 2" "128 1" "128 2"; do set -- $FS; python bench.py --steps 512 --warmup 2 --fuse $1 --streams $2 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/fuse_sweep.jsonl; done
# the second shape of cfg 2 (256 instances x 4096 cycles; 512 batches = 2048 waves fill the chip)
for A in "--fuse 64 --steps 128 --warmup 64" "--fuse 256 --steps 512 --warmup 256"; do python bench.py --no-cpu-baseline --instances 256 --cycles 4096 $A 2>/dev/null | grep '^{' >> $OUT/long_traces.jsonl; done
# the other configurations (parity cases rather than bench lines)
python bench.py --cfg 1 --commit-mask 0 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/other_cfgs.jsonl
python bench.py --cfg 4 --instances 4096 --cycles 1024 --steps 64 --warmup 16 --fuse 16 --commit-mask 0 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/other_cfgs.jsonl
python bench.py --cfg 4 --instances 4096 --cycles 1024 --steps 32 --warmup 16 --fuse 16 --commit-mask 7 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/other_cfgs.jsonl
python bench.py --cfg 2 --steps 64 --warmup 32 --fuse 32 --commit-mask 7 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/other_cfgs.jsonl
# BASELINE configs[3] (precompile-dominant, 512 instances = one GPU's share): its own bench line with roofline + cpu_baseline
# from the traced process, a lone batch, and the kernel stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg3 -o $TAG -- python bench.py --cfg 3 --commit-mask 0 --fuse 256 --steps 256 --warmup 256 --streams 1 > $OUT/trace_cfg3.log 2>&1; grep '^{' $OUT/trace_cfg3.log > $OUT/cfg3_bench.json
python bench.py --cfg 3 --commit-mask 0 --fuse 1 --steps 4 --warmup 2 --streams 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/cfg3_lone_batch.json
# ... the lone batch as a caller after latency runs it: 2 lanes per wave, keccak256 served by helper waves (second line of the file)
python bench.py --cfg 3 --commit-mask 0 --fuse 1 --steps 8 --warmup 2 --streams 1 --lanes 2 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/cfg3_lone_batch.json
python bench.py --cfg 3 --commit-mask 0 --fuse 16 --steps 32 --warmup 16 --streams 1 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/other_cfgs.jsonl
# the files the judge (and tests/test_bench_contract.py) read: copied under profiles/ with the tag
cp $OUT/bench.json profiles/${TAG}_bench.json; cp $OUT/bench_driver.json profiles/${TAG}_driver_bench.json; cp $OUT/bench_driver_full.json profiles/${TAG}_driver_full_line.json
cp $OUT/other_configs_traced.jsonl profiles/${TAG}_other_configs_traced.jsonl; for i in 0 1 2 3; do cp $OUT/oc${i}_kernel_stats.csv profiles/${TAG}_oc${i}_kernel_stats.csv; done
cp $OUT/bench_plain.json profiles/${TAG}_bench_plain.json; cp $OUT/bench_driver_plain.json profiles/${TAG}_driver_bench_plain.json
cp $OUT/trace/${TAG}_kernel_stats.csv profiles/${TAG}_kernel_stats.csv; cp $OUT/trace_driver/${TAG}_kernel_stats.csv profiles/${TAG}_driver_kernel_stats.csv
for f in instance_sweep fuse_sweep long_traces other_cfgs; do cp $OUT/$f.jsonl profiles/${TAG}_$f.jsonl; done
cp $OUT/occupancy_probe.txt profiles/${TAG}_occupancy_probe.txt
cp $OUT/cfg3_bench.json profiles/${TAG}_cfg3_line.json; cp $OUT/cfg3_lone_batch.json profiles/${TAG}_cfg3_lone_batch.json; cp $OUT/trace_cfg3/${TAG}_kernel_stats.csv profiles/${TAG}_cfg3_kernel_stats.csv
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_* gpurun_out/profiles_$TAG/   # profiles/ itself does not travel back: gpurun_out/ does
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, json, collections, os
out, tag = sys.argv[1], sys.argv[2]
res = {}
for key, d in (("kernel_stats_driver", "trace_driver"), ("kernel_stats_default", "trace")):
    for f in glob.glob(os.path.join(out, d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "zkw_" in r.get("Name", ""):
                res.setdefault(key, {})[r["Name"].split("(")[0]] = r
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
res["pmc_command"] = "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline (one launch = 20 batches = 1280 waves x 256 cycles)"
res["pmc_avg_per_dispatch"] = {k: sum(v) / len(v) for k, v in pm.items()}
res["pmc_dispatches"] = {k: len(v) for k, v in pm.items()}
wc = 20 * 64 * 256.0
res["pmc_per_wave_cycle"] = {k: sum(v) / len(v) / wc for k, v in pm.items() if k.startswith("SQ_")}
if "WRITE_SIZE" in pm and "FETCH_SIZE" in pm:
    w = sum(pm["WRITE_SIZE"]) / len(pm["WRITE_SIZE"]) * 1024.0
    f = 2.0 * sum(pm["FETCH_SIZE"]) / len(pm["FETCH_SIZE"]) * 1024.0  # FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM)
    import subprocess
    res["traffic"] = {"fused_batches": 20, "write_bytes": w, "fetch_bytes_corrected_x2": f, "hbm_bytes_per_launch": w + f, "hbm_bytes_per_vm_cycle": (w + f) / (20 * 4096 * 256.0),
                      "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on the driver's command, collection " + tag,
                      "kernel_source_sha256": subprocess.check_output([sys.executable, "bench.py", "--kernel-source-hash"], text=True).strip()}
    json.dump(res["traffic"], open(os.path.join("gpurun_out", "profiles_" + tag, "traffic.json"), "w"), indent=1)  # -> profiles/traffic.json (bench.py: roofline.traffic)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
json.dump(res, open(os.path.join("gpurun_out", "profiles_" + tag, tag + "_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3500])
PY
