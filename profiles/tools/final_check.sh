# end-of-round check on the GPU box: smoke(), the single-rank collective path of bench.py (RCCL, world size 1), default bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python bench.py --gpus 1 --force-collective --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 64 --warmup 32 --no-cpu-baseline --force-collective 2>&1 | tail -1 | cut -c1-400
python bench.py 2>&1 | tail -1 | cut -c1-400
