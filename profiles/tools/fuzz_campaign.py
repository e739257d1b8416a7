"""One-off differential campaign: fuzz tapes (synth.fuzz_workload) through libzkw.so and the oracle for many seeds, lane
widths and tape lengths, also with every light group forced onto the variant-group path, commitments included.
   python profiles/tools/fuzz_campaign.py <first seed> <n seeds> [uniform]     (prints one line per seed; exits 1 on a mismatch)
`uniform` (round 5): synth.uniform_fuzz instead — ONE random tape per workload, per-instance data: the waves stay at one pc and the
cycle kernel's short cycle executes what qualifies; every third seed with the short cycle switched off (test hook)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
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
UNIFORM = len(sys.argv) > 3 and sys.argv[3] == "uniform"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import _metamorphic as M  # noqa: E402
# round 4: every fourth seed under a renumbered, every fourth under an estranged ISA table (tests/_metamorphic.py)
TABLES = [("default", K.Isa()), ("renumbered", M.renumbered(0x7AB1E)), ("estranged", M.estranged(0xE57A))]
CTX = {name: (isa_, _campaign_backend().open(isa_), load_oracle().open(isa_)) for name, isa_ in TABLES}
bad = 0
t0 = time.time()
for k in range(count):
    seed = first + k
    lanes = (64, 64, 16, 0, 8, 1)[k % 6]
    n_ops = (96, 160, 64, 128)[k % 4]
    forced = (k % 3) == 2
    table = ("default", "renumbered", "default", "estranged")[k % 4]
    isa, prod, orc = CTX[table]
    prod.set_option(K.OPT_DEBUG_FLAGS, (1 << 24) if forced else 0)
    wl = synth.uniform_fuzz(isa, n_instances=(320, 512, 192)[k % 3], n_ops=2 * n_ops, seed=seed) if UNIFORM else synth.fuzz_workload(isa, n_instances=512, n_ops=n_ops, seed=seed)
    bo = orc.create_batch(wl); bo.reset(); bo.run(wl.n_cycles); bo.sync()
    wl.limits["lanes_per_wave"] = lanes
    bp = prod.create_batch(wl); bp.reset(); bp.run(wl.n_cycles); bp.sync()
    limited = compared = executed = 0
    msg = ""
    for i in range(wl.n_instances):
        tp = bp.trace(i)
        if int(tp["status"]) == K.STATUS_LIMIT:
            limited += 1
            continue
        ok, why = K.traces_equal(bo.trace(i), tp)
        if not ok:
            bad += 1
            msg = "MISMATCH instance %d: %s" % (i, why[:120])
            break
        compared += 1
        executed += len(tp["records"])
    if not msg:
        co, cp = bo.commitments(), bp.commitments()
        keep = np.array([int(bp.trace(i)["status"]) != K.STATUS_LIMIT for i in range(wl.n_instances)])
        if not np.array_equal(co[keep], cp[keep]):
            bad += 1
            msg = "COMMITMENT MISMATCH"
    print("seed %#x table %-10s lanes %2d ops %3d forced %d: compared %d limited %d cycles %d %s" % (seed, table, lanes, n_ops, forced, compared, limited, executed, msg), flush=True)
    bo.destroy(); bp.destroy()
print("done: %d seeds, %d bad, %.0f s" % (count, bad, time.time() - t0))
sys.exit(1 if bad else 0)
