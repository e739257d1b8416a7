# Memory-system counters of the cycle kernel on the driver's command (round 5): is the launch held by the path to HBM?
# (every pass under `timeout`: a pass with the *_LEVEL counters aborted at start-up and then sat until gpurun's limit)
# Each group of counters in its own rocprofv3 run (PMC only: gpurun refuses --pmc mixed with tracing domains).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r08_mem_pmc; mkdir -p $OUT
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-host-legs --repeats 0 $EXTRA"
i=0
for C in "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_TOO_MANY_EA_WRREQS_STALL TCC_BUSY GRBM_GUI_ACTIVE" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM" \
         "TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TA_BUSY" \
         "TCC_HIT TCC_MISS TCC_REQ TCC_WRITE TCC_WRITEBACK TCC_TAG_STALL TCC_IB_STALL"; do
  timeout 150 rocprofv3 --pmc $C --kernel-include-regex zkw_cycle_kernel --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
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
