#!/bin/bash
# permutation throughput + the commitment-bound configurations + the driver's command, after a change of zkw_goldilocks.hip.h
set -u
OUT=gpurun_out/${1:-perm_ab}; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -I era-zk_evm_amd/csrc profiles/tools/perm_probe.hip -o /tmp/perm_probe 2>/dev/null && /tmp/perm_probe > $OUT/perm_probe.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu -k "commit or sponge or golden or fuzz or smoke or reset" > $OUT/pytest_commit.txt 2>&1; tail -3 $OUT/pytest_commit.txt
python bench.py --cfg 4 --instances 4096 --cycles 1024 --steps 32 --warmup 16 --fuse 16 --commit-mask 7 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/cfg4_mask7.json
python bench.py --cfg 2 --steps 64 --warmup 32 --fuse 32 --commit-mask 7 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/cfg2_mask7.json
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/driver.jsonl; done
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/default.json
cat $OUT/perm_probe.txt
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+'/*.json*')):
    for l in open(f):
        d=json.loads(l); print(f.split('/')[-1], round(d['value']/1e9,3), 'G', d['roofline']['frac'])
PY
