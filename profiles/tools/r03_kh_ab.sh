#!/bin/bash
# same-box A/B of keccak-helper variants on the cfg-3 lone batch (2 lanes per wave): ab_<name>.so copied over libzkw.so in turn
OUT=gpurun_out/$1; mkdir -p $OUT; shift
cp era-zk_evm_amd/libzkw.so /tmp/libzkw_keep.so
for round in 1 2 3; do
for name in "$@"; do
  cp era-zk_evm_amd/ab_$name.so era-zk_evm_amd/libzkw.so
  timeout 300 python bench.py --cfg 3 --commit-mask 0 --fuse 1 --steps 8 --warmup 2 --streams 1 --lanes 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$name ms_per_step', round(d['ms_per_step'],4))" | tee -a $OUT/ab.txt
done
done
cp /tmp/libzkw_keep.so era-zk_evm_amd/libzkw.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "helper" 2>&1 | tail -2
