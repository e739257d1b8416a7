"""One-off differential campaign for the arena: random plans of sequential far calls (callee K keeps its heap alive as
returndata, P panics, N forwards a nested callee's returndata) under small max_far_frames, through libzkw.so and the oracle:
traces, and every page a run touched read back through zkw_batch_get_page (dump_page_content_as_u256_words).  Plans whose
live returndata pages exceed the slots must stop with ZKW_STATUS_LIMIT in the product (the reference keeps them all).
   python profiles/tools/far_call_campaign.py <first seed> <n seeds>"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
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
from test_emu_parity import compare_pages  # noqa: E402

first, count = int(sys.argv[1], 0), int(sys.argv[2])
isa = K.Isa()
prod = _campaign_backend().open(isa)
orc = load_oracle().open(isa)
bad = 0
t0 = time.time()
for k in range(count):
    seed = first + k
    rng = random.Random(seed)
    F = rng.choice([3, 4, 5, 6])
    n = rng.randrange(8, 48)
    plan = "".join(rng.choice("PPPPKN") for _ in range(n))
    # live returndata pages at the end: one per K, one per N (its inner K), all owned by the bootloader frame
    need = 1 + sum(1 for c in plan if c in "KN") + (1 if "N" in plan else 0)
    # ... and while a call runs its own slot (and the nested callee's, for N) sits beside the pages kept so far: a plan
    # that calls again behind its last K needs one slot more than its kept pages (seeds 0x583d / 0x5841 of the round-4 long
    # campaign stopped on the limit there, rightly, under the first formula alone)
    kept = 0
    for c in plan:
        need = max(need, 1 + kept + 1 + (1 if c == "N" else 0))
        if c in "KN":
            kept += 1
    lanes = (0, 64, 8, 1)[k % 4]
    wl = synth.many_far_calls(isa, n_calls=n, plan=plan, n_instances=70, seed=seed, max_far_frames=F)
    bo = orc.create_batch(wl); bo.reset(); bo.run(wl.n_cycles); bo.sync()
    wl.limits["lanes_per_wave"] = lanes
    bp = prod.create_batch(wl); bp.reset(); bp.run(wl.n_cycles); bp.sync()
    msg, limited = "ok", 0
    for i in range(wl.n_instances):
        tp, to = bp.trace(i), bo.trace(i)
        if tp["status"] == K.STATUS_LIMIT:
            limited += 1
            continue
        ok, why = K.traces_equal(to, tp)
        if not ok:
            msg = "MISMATCH instance %d: %s" % (i, why[:100]); bad += 1
            break
    if msg == "ok":
        if limited and need <= F:
            msg = "UNEXPECTED LIMIT (%d instances) need %d" % (limited, need); bad += 1
        elif not limited:
            try:
                compare_pages(bo, bp, wl, [0, 33, 69], n_words=24)
            except AssertionError as e:
                msg = "PAGE MISMATCH %s" % str(e)[:100]; bad += 1
    print("seed %#x F %d lanes %2d plan %-48s need %2d limited %2d %s" % (seed, F, lanes, plan, need, limited, msg), flush=True)
    bo.destroy(); bp.destroy()
print("done: %d seeds, %d bad, %.0f s" % (count, bad, time.time() - t0))
sys.exit(1 if bad else 0)
