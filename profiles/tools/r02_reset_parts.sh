# duration of the reset kernel in the driver's command with parts of it left out (ZKW_RESET_SKIP: 1 flat copies, 2 storage
# slots, 4 heap words, 8 commitment tails); the run is then wrong — timing only.   usage: r02_reset_parts.sh <tag>
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=$R/gpurun_out/$1; mkdir -p $T
cd /tmp
for SK in 0 1 2 4 8 15; do
  ZKW_RESET_SKIP=$SK rocprofv3 --kernel-trace --stats --output-format csv -d $T/sk$SK -o x -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $T/sk$SK.log 2>&1
  python3 - $T/sk$SK $SK <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_reset" in r["Name"]: print("skip", sys.argv[2], "reset kernel avg us", float(r["AverageNs"]) / 1e3, "calls", r["Calls"])
PY
done
