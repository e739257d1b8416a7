"""Round 6 (no GPU): the BASELINE configurations at FULL size through the 64-lane CPU emulation of the kernels (tests/emu:
libzkw_emu64.so), EVERY instance compared with the oracle record for record — the `-m gpu` tests of the same sizes sample every
37th / 211th instance (test_cfg2_full_size_sampled, test_cfg4_full_size_sampled) — plus all three queue commitments.
   python profiles/tools/r10_full_size_emu64.py [cfg ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import numpy as np  # noqa: E402
import era_zk_evm_amd  # noqa: E402,F401
from era_zk_evm_amd import capi as K, synth  # noqa: E402
from _oracle import load_oracle  # noqa: E402
import build_emu  # noqa: E402

isa = K.Isa()
orc = load_oracle().open(isa)
emu = K.Backend(build_emu.build(wave=64), "zkw_").open(isa)
CASES = {1: lambda: synth.make(1, isa, n_instances=4096), 2: lambda: synth.make(2, isa), 3: lambda: synth.make(3, isa, n_instances=512),
         4: lambda: synth.make(4, isa, n_instances=4096, n_cycles=1024), 22: lambda: synth.make(2, isa, n_instances=256, n_cycles=4096)}
bad = 0
for cfg in [int(a) for a in sys.argv[1:]] or [1, 2, 22, 3, 4]:
    wl = CASES[cfg]()
    t0 = time.time()
    bo = orc.create_batch(wl); bo.reset(); bo.run(wl.n_cycles); bo.sync()
    t1 = time.time()
    bp = emu.create_batch(wl); bp.reset(); bp.run(wl.n_cycles); bp.sync()
    t2 = time.time()
    cycles = int(bp.stats()["cycles"])
    mism = 0
    for i in range(wl.n_instances):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        if not ok:
            mism += 1
            print("  MISMATCH cfg %d instance %d: %s" % (cfg, i, why[:100]))
    same_c = bool(np.array_equal(bo.commitments(), bp.commitments()))
    print("cfg %2d %s: %d instances x %d cycles = %d VM cycles, EVERY instance compared: %d mismatches; commitments equal: %s  (oracle %.0f s, emulation %.0f s, compare %.0f s)"
          % (cfg, wl.name, wl.n_instances, wl.n_cycles, cycles, mism, same_c, t1 - t0, t2 - t1, time.time() - t2), flush=True)
    bad += mism + (0 if same_c else 1)
    bo.destroy(); bp.destroy()
sys.exit(1 if bad else 0)
