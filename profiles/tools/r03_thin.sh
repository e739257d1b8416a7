#!/bin/bash
# lone batches of 512 instances at several lane widths: is a thin-wave geometry the better default for small batches?
OUT=gpurun_out/$1; mkdir -p $OUT
for cfg in 2 4 1 3; do
for lanes in 0 2 8 16; do
  timeout 300 python bench.py --cfg $cfg --instances 512 --commit-mask 0 --fuse 1 --steps 8 --warmup 2 --streams 1 --lanes $lanes --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('cfg $cfg lanes $lanes ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['kernel_ms'],4))" | tee -a $OUT/thin.txt
done
done
