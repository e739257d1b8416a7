"""Debug aid: a wave of 64 different arithmetic tapes (variant grouping) against the oracle; prints the first differing
cycle of the first differing instances with the instruction that ran in it."""
import sys
sys.path.insert(0, '.')
import numpy as np
import era_zk_evm_amd
from era_zk_evm_amd import capi as K, synth
from tests._oracle import load_oracle
isa = K.Isa()
prod = K.load_product().open(isa)
orc = load_oracle().open(isa)
n = 64
wl = synth.make(1, isa, n_instances=n, n_cycles=64)
tapes = [None]
for i in range(1, n):
    tp = synth.arith_tape(isa, 64, synth.ScalarRng(1000 + i))
    tapes.append(tp)
    wl.blobs.append(K.pack_code(tp))
    wl.code_pages.append((i, 1, synth.BOOTLOADER_CODE_PAGE, len(wl.blobs) - 1))
wl.limits["lanes_per_wave"] = 64
bo = orc.create_batch(wl); bo.reset(); bo.run(wl.n_cycles)
bp = prod.create_batch(wl); bp.reset(); bp.run(wl.n_cycles); bp.sync()
bad = 0
for i in range(n):
    a, b = bo.trace(i), bp.trace(i)
    ok, why = K.traces_equal(a, b)
    if ok:
        continue
    bad += 1
    if bad > 4:
        continue
    ra, rb = a["records"], b["records"]
    for k in range(min(len(ra), len(rb))):
        if ra[k].tobytes() != rb[k].tobytes():
            regs_a, regs_b = ra[k]["registers"], rb[k]["registers"]
            diff = [r + 1 for r in range(15) if regs_a[r].tobytes() != regs_b[r].tobytes()]
            pc_before = int(ra[k - 1]["tail"]["pc"]) if k else 0
            code = np.frombuffer(wl.blobs[[cp for cp in wl.code_pages if cp[0] == i][-1][3] if i else 0].tobytes() if hasattr(wl.blobs[0], 'tobytes') else b'', dtype='<u8') if False else None
            print("instance", i, "cycle", k, "pc before", pc_before, "regs differing", diff, "tail equal", ra[k]["tail"].tobytes() == rb[k]["tail"].tobytes())
            print("  oracle ", [hex(int(x)) for x in regs_a[diff[0] - 1]] if diff else None)
            print("  product", [hex(int(x)) for x in regs_b[diff[0] - 1]] if diff else None)
            if k:
                print("  previous value", [hex(int(x)) for x in ra[k - 1]["registers"][diff[0] - 1]] if diff else None)
            break
print("instances differing:", bad, "of", n)
