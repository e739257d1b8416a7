# Average latencies of the cycle kernel's memory instructions on the driver's command (derived rocprofv3 metrics), round 5
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r08_lat_pmc; mkdir -p $OUT
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-host-legs --repeats 0 $EXTRA"
i=0
for C in "VmemLatency" "LdsLatency SmemLatency" "InstrFetchLatency" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"; do
  timeout 150 rocprofv3 --pmc $C --kernel-include-regex zkw_cycle_kernel --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i ($C): failed / timed out"
  i=$((i+1))
done
python - $OUT <<'PY'
import csv, glob, sys, os, collections
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "zkw_cycle_kernel" in r.get("Kernel_Name", ""):
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(pm):
    v = pm[k]
    print("%-36s n=%4d avg per dispatch %.6g" % (k, len(v), sum(v) / len(v)))
PY
