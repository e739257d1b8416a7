"""Generates tests/golden/trace_digests.json from the oracle (run: python tests/golden/make_golden.py)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def digest_case(oracle, wl):
    from era_zk_evm_amd import capi as K
    b = oracle.create_batch(wl)
    b.reset()
    b.run(wl.n_cycles)
    b.sync()
    h = hashlib.sha256()
    for i in range(wl.n_instances):
        t = b.trace(i)
        h.update(bytes([t["status"]]))
        for k in K.TRACE_ARRAYS:
            h.update(t[k].tobytes())
        h.update(t["final_state"].tobytes())
    b.destroy()
    return h.hexdigest()


if __name__ == "__main__":
    from era_zk_evm_amd import capi as K
    from test_emu_parity import CASES
    isa = K.Isa()
    orc = K.load_oracle().open(isa)
    out = {name: digest_case(orc, CASES[name](isa)) for name in sorted(CASES)}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "trace_digests.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))
