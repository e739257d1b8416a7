"""Where the time of a lone cfg-3 batch goes: the whole workload, the keccak calls alone, the sha256 calls alone (kernel time of
one launch, best of 5).   python profiles/tools/cfg3_parts.py <lanes>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from era_zk_evm_amd import capi as K, synth  # noqa: E402

lanes = int(sys.argv[1])
isa = K.Isa()
prod = K.load_product().open(isa)
for label, kw in (("whole", {}), ("keccak", dict(sha_rounds=(1, 1, 1, 1))), ("sha256", dict(keccak_k=(1, 1, 1, 1))), ("neither", dict(keccak_k=(1, 1, 1, 1), sha_rounds=(1, 1, 1, 1)))):
    wl = synth.make(3, isa, n_instances=512, **kw)
    wl.limits["lanes_per_wave"] = lanes
    b = prod.create_batch(wl)
    best = None
    for _ in range(5):
        b.reset(); b.run(wl.n_cycles); b.sync()
        ms = float(b.stats()["kernel_ms"])
        best = ms if best is None else min(best, ms)
    print("lanes %d %-8s kernel %.3f ms" % (lanes, label, best), flush=True)
    b.destroy()
