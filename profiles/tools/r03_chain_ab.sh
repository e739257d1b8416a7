# chain kernel A/B (ab_*.so): cfg 2 / cfg 4 with all three commitments, 3 rounds + per-kernel stats of one run each
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
P='import sys,json; [print("%s fuse=%d mask=%d ms_per_step=%.4f kernel_ms=%.3f cycles/s=%.4g"%(j["config"]["workload"], j["config"]["batches_per_fused_launch"], j["config"]["commit_mask"], j["ms_per_step"], j["kernel_ms"], j["value"])) for j in map(json.loads, sys.stdin)]'
cp era-zk_evm_amd/libzkw.so /tmp/keep.so
for R in 1 2 3; do
for L in era-zk_evm_amd/ab_*.so; do
  cp $L era-zk_evm_amd/libzkw.so; echo "== $L" | tee -a $T/commit7.txt
  python bench.py --no-cpu-baseline --cfg 4 --cycles 1024 --commit-mask 7 --fuse 16 --steps 32 --warmup 16 2>&1 | grep '^{\|rror' | python -c "$P" | tee -a $T/commit7.txt
  python bench.py --no-cpu-baseline --cfg 2 --commit-mask 7 --fuse 32 --steps 64 --warmup 32 2>&1 | grep '^{\|rror' | python -c "$P" | tee -a $T/commit7.txt
done
done
for L in era-zk_evm_amd/ab_*.so; do
  cp $L era-zk_evm_amd/libzkw.so; N=$(basename $L .so)
  rocprofv3 --kernel-trace --stats --output-format csv -d $T/prof_$N -o x -- python bench.py --no-cpu-baseline --cfg 2 --commit-mask 7 --fuse 32 --steps 64 --warmup 32 > /dev/null 2>&1
  echo "== $N"; python - $T/prof_$N <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:40], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
done
cp /tmp/keep.so era-zk_evm_amd/libzkw.so
