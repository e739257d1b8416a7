#!/bin/bash
# keccak helper waves: parity campaign at thin waves, cfg-3 lone batch at several lane widths
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 300 python profiles/tools/dbg_commit_seed.py 0x400c 64 96 > $OUT/dbg_seed.txt 2>&1
timeout 600 python - > $OUT/kh_parity.txt 2>&1 <<'PY'
import sys, random, time
sys.path.insert(0, ".")
from era_zk_evm_amd import capi as K, synth
from tests._oracle import load_oracle
isa = K.Isa(); prod = K.load_product().open(isa); orc = load_oracle().open(isa)
bad = 0
for k in range(36):
    seed = 0x5100 + k
    rng = random.Random(seed)
    kb = tuple(rng.choice([0, 1, 31, 32, 33, 135, 136, 137, 200, 271, 272, 273, rng.randrange(0, 700)]) for _ in range(4))
    ku = tuple(rng.randrange(0, 32) for _ in range(4))
    sr = tuple(rng.choice([1, 2, 3, 5]) for _ in range(4))
    lanes = (2, 8, 1, 4, 3, 5)[k % 6]
    wl = synth.make(3, isa, n_instances=(96, 40, 7)[k % 3], seed=seed, keccak_bytes=kb, keccak_unalign=ku, sha_rounds=sr)
    bo = orc.create_batch(wl); bo.reset(); bo.run(wl.n_cycles); bo.sync()
    wl.limits["lanes_per_wave"] = lanes
    bp = prod.create_batch(wl)
    if k % 2:
        bp.reset(); bp.run(wl.n_cycles)
    else:
        prod.step_many([bp], wl.n_cycles, 7)
    bp.sync()
    if k % 2 == 0:
        import numpy as np
        if not np.array_equal(bo.commitments(), bp.commitments()): print("  COMMITMENT MISMATCH"); bad += 1
    msg = ""
    for i in range(wl.n_instances):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        if not ok:
            bad += 1; msg = "MISMATCH instance %d: %s" % (i, why[:160]); break
    print("seed %#x lanes %d n %d keccak %s unalign %s %s" % (seed, lanes, wl.n_instances, kb, ku, msg or "ok"), flush=True)
    bo.destroy(); bp.destroy()
print("done, %d bad" % bad)
PY
for lanes in 2 8 0; do
  for flags in 0 268435456; do
    echo "lanes $lanes debug_flags $flags" >> $OUT/lone.txt
    ZKW_DEBUG_FLAGS=$flags timeout 300 python bench.py --cfg 3 --commit-mask 0 --fuse 1 --steps 6 --warmup 2 --streams 1 --lanes $lanes --no-cpu-baseline 2>>$OUT/lone.err | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  ms_per_step', d['ms_per_step'], 'value', d['value'], 'lone_launch_ms', d['roofline'].get('lone_launch_ms'))" >> $OUT/lone.txt
  done
done
cat $OUT/dbg_seed.txt; tail -5 $OUT/kh_parity.txt; cat $OUT/lone.txt
