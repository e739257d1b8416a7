"""Differential look at one fuzz seed whose traces match but whose commitments do not: which instance, which queue.
   python profiles/tools/dbg_commit_seed.py <seed> <lanes> <n_ops>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from era_zk_evm_amd import capi as K, synth  # noqa: E402
from tests._oracle import load_oracle  # noqa: E402

seed, lanes, n_ops = int(sys.argv[1], 0), int(sys.argv[2]), int(sys.argv[3])
isa = K.Isa()
prod = K.load_product().open(isa)
orc = load_oracle().open(isa)
wl = synth.fuzz_workload(isa, n_instances=512, n_ops=n_ops, seed=seed)
bo = orc.create_batch(wl); bo.reset(); bo.run(wl.n_cycles); bo.sync()
wl.limits["lanes_per_wave"] = lanes
bp = prod.create_batch(wl); bp.reset(); bp.run(wl.n_cycles); bp.sync()
co, cp = bo.commitments(), bp.commitments()
st = np.array([int(bp.trace(i)["status"]) for i in range(wl.n_instances)])
keep = st != K.STATUS_LIMIT
d = np.argwhere((co != cp).any(axis=-1) & keep[:, None])
print("limited:", np.nonzero(~keep)[0].tolist())
print("mismatching (instance, queue):", d.tolist())
for i, qn in d[:8]:
    i = int(i)
    t = bp.trace(i)
    print("instance", i, "queue", int(qn), "status", int(t["status"]), "records", len(t["records"]), "mem", len(t["mem"]) if "mem" in t else "?",
          "keys", [k for k in t.keys()][:12])
    L = lanes or 64
    w = i // L
    print("  wave", w, "statuses of its lanes:", st[w * L:(w + 1) * L].tolist())
# a second run of the same batch: is it deterministic?
bp.reset(); bp.run(wl.n_cycles); bp.sync()
cp2 = bp.commitments()
print("second run equal to first:", np.array_equal(cp, cp2), " equal to oracle on kept:", np.array_equal(co[keep], cp2[keep]))
