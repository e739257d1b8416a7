#!/bin/bash
# The shortest useful GPU call (when little of the round is left): is HEAD green on an MI355X, and the driver's command.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/tools/r10_quick_gpu_call.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r10_quick; mkdir -p $OUT
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $OUT/pytest.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.log 2>&1; grep '^{' $OUT/bench_full.log > $OUT/driver_full_line.json
tail -3 $OUT/pytest.log; tail -1 $OUT/smoke.log; cut -c1-600 $OUT/driver_full_line.json
