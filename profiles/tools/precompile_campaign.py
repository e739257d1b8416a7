"""One-off differential campaign for the precompile path: cfg-3 workloads (sha256 over the frame's own heap, keccak256
through a fat pointer into the caller's heap) with random message lengths, byte misalignments and round counts, through
libzkw.so and the oracle — traces (incl. the precompile-tagged memory queries) and the digests written to the heaps.
   python profiles/tools/precompile_campaign.py <first seed> <n seeds>"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from era_zk_evm_amd import capi as K, synth  # noqa: E402


def _campaign_backend():
    """the product (libzkw.so on a GPU) — or, with ZKW_CAMPAIGN_BACKEND=emu64 / emu1, the same sources compiled for the CPU
    (tests/emu: 64-lane waves on the SIMT engine / one-lane waves): the campaigns then run without a GPU"""
    which = os.environ.get("ZKW_CAMPAIGN_BACKEND", "")
    if which in ("emu64", "emu1"):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "emu"))
        import build_emu
        defs = tuple(d for d in os.environ.get("ZKW_CAMPAIGN_DEFINES", "").split(",") if d)  # (an A/B partner: e.g. ZKW_SHORT_CLASS,ZKW_SHORT_STACK)
        return K.Backend(build_emu.build(wave=64 if which == "emu64" else 1, defines=defs, tag="_".join(d.lower() for d in defs)), "zkw_")
    return K.load_product()

from tests._oracle import load_oracle  # noqa: E402

first, count = int(sys.argv[1], 0), int(sys.argv[2])
isa = K.Isa()
prod = _campaign_backend().open(isa)
orc = load_oracle().open(isa)
bad = 0
t0 = time.time()
for k in range(count):
    seed = first + k
    rng = random.Random(seed)
    kb = tuple(rng.choice([0, 1, 31, 32, 33, 135, 136, 137, 200, 271, 272, 273, rng.randrange(0, 700)]) for _ in range(4))
    ku = tuple(rng.randrange(0, 32) for _ in range(4))
    sr = tuple(rng.choice([1, 2, 3, 5, 8, rng.randrange(1, 20)]) for _ in range(4))
    lanes = (0, 64, 16, 4, 2, 8, 1)[k % 7]  # (<= 8: keccak256 served by helper waves)
    wl = synth.make(3, isa, n_instances=96, seed=seed, keccak_bytes=kb, keccak_unalign=ku, sha_rounds=sr)
    bo = orc.create_batch(wl); bo.reset(); bo.run(wl.n_cycles); bo.sync()
    wl.limits["lanes_per_wave"] = lanes
    bp = prod.create_batch(wl); bp.reset(); bp.run(wl.n_cycles); bp.sync()
    msg = ""
    for i in range(wl.n_instances):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        if not ok:
            bad += 1
            msg = "MISMATCH instance %d: %s" % (i, why[:120])
            break
    print("seed %#x lanes %2d keccak bytes %s unalign %s sha rounds %s %s" % (seed, lanes, kb, ku, sr, msg or "ok"), flush=True)
    bo.destroy(); bp.destroy()
print("done: %d seeds, %d bad, %.0f s" % (count, bad, time.time() - t0))
sys.exit(1 if bad else 0)
