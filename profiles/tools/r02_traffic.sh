# HBM traffic of the cycle kernel (rocprofv3 PMC, separate passes) and its attribution by ablation:
#   ZKW_DEBUG_FLAGS 0 = full kernel, 1 = no CycleRecord stores (tails + deltas), 2 = no query-stream stores, 3 = neither
# usage: r02_traffic.sh <tag> [fuse]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
F=${2:-32}
CMD="python bench.py --fuse $F --streams 1 --steps $((F*2)) --warmup $F --no-cpu-baseline"
for FL in 0 1 2 3; do
  ZKW_DEBUG_FLAGS=$FL rocprofv3 --pmc WRITE_SIZE --output-format csv -d $T/w$FL -o x -- $CMD > $T/w$FL.log 2>&1
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/f0 -o x -- $CMD > $T/f0.log 2>&1
python - $T $F <<'PY'
import csv, glob, sys, os, json
out, F = sys.argv[1], int(sys.argv[2])
cycles = F * 4096 * 256.0
res = {}
for d in ("w0", "w1", "w2", "w3", "f0"):
    vals = []
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    if vals:
        vals.sort()
        med = vals[len(vals) // 2]
        res[d] = {"dispatches": len(vals), "median_counter": med}
# WRITE_SIZE / FETCH_SIZE are reported in KB by rocprofv3; FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md, HBM)
def kb(d):
    return res[d]["median_counter"] * 1024.0 if d in res else None
summary = {"fused_batches": F, "cycles_per_launch": cycles, "raw": res}
if "w0" in res:
    summary["write_bytes_per_cycle"] = {k: kb(k) / cycles for k in ("w0", "w1", "w2", "w3") if k in res}
if "f0" in res:
    summary["fetch_bytes_per_cycle_doubled"] = 2.0 * kb("f0") / cycles
    if "w0" in res:
        summary["hbm_bytes_per_launch"] = kb("w0") + 2.0 * kb("f0")
        summary["hbm_bytes_per_cycle"] = summary["hbm_bytes_per_launch"] / cycles
# kernel time of each ablation (HIP events inside bench.py; profiled runs, so only their ratios matter)
km = {}
for fl in "0123":
    try:
        for ln in open(os.path.join(out, "w%s.log" % fl)):
            if ln.startswith("{"):
                km["w" + fl] = json.loads(ln)["kernel_ms"]
    except OSError:
        pass
summary["kernel_ms_by_ablation"] = km
json.dump(summary, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
