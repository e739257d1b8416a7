#!/bin/bash
# keccak helper waves, second pass: the new GPU tests + the commitment regression, then the cfg-3 lone batch
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "helper or nominal_share or cfg3 or keccak or fuzz or commit" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
for lanes in 2 4 8 0; do
    echo "lanes $lanes" >> $OUT/lone.txt
    timeout 300 python bench.py --cfg 3 --commit-mask 0 --fuse 1 --steps 6 --warmup 2 --streams 1 --lanes $lanes --no-cpu-baseline 2>>$OUT/lone.err | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  ms_per_step', d['ms_per_step'], 'value', d['value'], 'lone_launch_ms', d['roofline'].get('lone_launch_ms'))" >> $OUT/lone.txt
done
cat $OUT/lone.txt
