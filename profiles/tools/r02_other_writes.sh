# What the HBM writes of the cycle kernel are besides the witness streams: WRITE_SIZE with the record and query stores
# off (ZKW_DEBUG_FLAGS 3) and, on top, the heap-dirty atomics (32), the heap word stores (64), the stack stores (128) off.
# usage: r02_other_writes.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
F=20
CMD="python bench.py --fuse $F --streams 1 --steps $((F*2)) --warmup $F --no-cpu-baseline"
for FL in 0 3 35 99 227; do
  ZKW_DEBUG_FLAGS=$FL rocprofv3 --pmc WRITE_SIZE --output-format csv -d $T/w$FL -o x -- $CMD > $T/w$FL.log 2>&1
done
python - $T $F <<'PY'
import csv, glob, sys, os
out, F = sys.argv[1], int(sys.argv[2])
cycles = F * 4096 * 256.0
for d in ("w0", "w3", "w35", "w99", "w227"):
    vals = []
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    if vals:
        vals.sort()
        print(d, "dispatches", len(vals), "write bytes per VM cycle %.1f" % (vals[len(vals) // 2] * 1024.0 / cycles))
PY
