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


def net_state_digest(oracle, wl):
    """SHA-256 over the oracle's final net states (zkw_net_state arrays) of every instance"""
    b = oracle.create_batch(wl)
    b.reset()
    b.run(wl.n_cycles)
    b.sync()
    h = hashlib.sha256()
    for i in range(wl.n_instances):
        ns = b.net_state(i)
        for k in ("storage_history", "event_history", "events", "l1_messages", "final_storage"):
            h.update(len(ns[k]).to_bytes(4, "little"))
            h.update(ns[k].tobytes())
    b.destroy()
    return h.hexdigest()


def net_state_cases(isa):
    from era_zk_evm_amd import capi as K, synth
    cases = {"cfg4_3": synth.make(4, isa, n_instances=3)}
    for outer in (K.RET_OK, K.RET_REVERT, K.RET_PANIC):
        for inner in (K.RET_OK, K.RET_PANIC):
            cases["nested_%d_%d" % (outer, inner)] = synth.nested_frames(isa, outer=outer, inner=inner)
    cases["nested_main_panics"] = synth.nested_frames(isa, main_panics=True)
    return cases


if __name__ == "__main__":
    from era_zk_evm_amd import capi as K
    from test_emu_parity import CASES
    isa = K.Isa()
    from _oracle import load_oracle

    orc = load_oracle().open(isa)
    out = {name: digest_case(orc, CASES[name](isa)) for name in sorted(CASES)}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "trace_digests.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))
    ns = {name: net_state_digest(orc, wl) for name, wl in sorted(net_state_cases(isa).items())}
    json.dump(ns, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "net_state_digests.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(ns, indent=1))
