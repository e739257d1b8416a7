# end-of-round check on the GPU box: smoke(), driver-like bench invocations, the single-rank RCCL path
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
P='import sys,json; j=json.loads(sys.stdin.read()); print("steps=%d warmup=%d fuse=%d groups=%d value=%.4g ms_per_step=%.4f frac=%.3f cpu=%s"%(j["steps"], j["warmup"], j["config"]["batches_per_fused_launch"], j["config"]["fused_groups_in_flight"], j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("cpu_baseline",{}).get("value")))'
for A in "" "--gpus 1 --steps 10 --warmup 3" "--gpus 1 --steps 1000 --warmup 10 --no-cpu-baseline" "--gpus 1 --steps 1 --warmup 0 --no-cpu-baseline"; do
  python bench.py $A 2> /tmp/err.txt | tail -1 | python -c "$P"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --no-cpu-baseline --force-collective 2> /tmp/err.txt | tail -1 | python -c "$P"
