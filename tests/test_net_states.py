"""Final net states: zkw_batch_get_net_state vs the oracle's restatement of get_final_net_states
(testing/mod.rs:42-71, testing/storage.rs:34-76, reference_impls/event_sink.rs:66-131) — CPU suite: the product
sources compiled by the emulation build (tests/emu)."""
import os
import sys

import numpy as np
import pytest

from era_zk_evm_amd import capi as K, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def emu(isa):
    import build_emu
    be = K.Backend(build_emu.build(), "zkw_").open(isa)
    yield be
    be.close()


def run(backend, wl, cycles=None):
    b = backend.create_batch(wl)
    b.reset()
    b.run(cycles or wl.n_cycles)
    b.sync()
    return b


def compare(bo, be, n):
    for i in range(n):
        ok, why = K.net_states_equal(bo.net_state(i), be.net_state(i))
        assert ok, (i, why)


@pytest.mark.parametrize("outer", [K.RET_OK, K.RET_REVERT, K.RET_PANIC])
@pytest.mark.parametrize("inner", [K.RET_OK, K.RET_REVERT, K.RET_PANIC])
def test_nested_frames(oracle, emu, isa, outer, inner):
    wl = synth.nested_frames(isa, outer=outer, inner=inner)
    bo, be = run(oracle, wl), run(emu, wl)
    compare(bo, be, wl.n_instances)
    ns = bo.net_state(0)
    rolled_st = int(((ns["storage_history"]["bools"] & K.LQ_ROLLBACK) != 0).sum())
    rolled_ev = int(((ns["event_history"]["bools"] & K.LQ_ROLLBACK) != 0).sum())
    # what the frame discipline implies (not only oracle == product): a failing outer frame takes everything of both
    # frames with it; a failing inner frame under a surviving outer one only its own write / event / L1 message
    if outer != K.RET_OK:
        # outer frame: writes key1, key1 again, key2 (3) + the inner write (1), which is rolled back twice if the inner
        # frame had already failed on its own; events: 2 own + (inner event + inner L1 message)
        assert (rolled_st, rolled_ev) == ((4, 4) if inner == K.RET_OK else (4, 4))
        assert len(ns["events"]) == 1 and len(ns["l1_messages"]) == 1  # only the main frame's own event + L1 message survive
    elif inner != K.RET_OK:
        assert (rolled_st, rolled_ev) == (1, 2)
        assert len(ns["events"]) == 3 and len(ns["l1_messages"]) == 1
    else:
        assert (rolled_st, rolled_ev) == (0, 0)
        assert len(ns["events"]) == 4 and len(ns["l1_messages"]) == 2
    # every rollback entry mirrors an earlier forward entry of the same slot
    sh = ns["storage_history"]
    for j in np.flatnonzero((sh["bools"] & K.LQ_ROLLBACK) != 0):
        earlier = [k for k in range(j) if sh[k]["timestamp"] == sh[j]["timestamp"] and not (sh[k]["bools"] & K.LQ_ROLLBACK)]
        assert earlier and sh[earlier[0]]["key"].tobytes() == sh[j]["key"].tobytes()


def test_main_frame_panic_rolls_everything_back(oracle, emu, isa):
    wl = synth.nested_frames(isa, main_panics=True)
    bo, be = run(oracle, wl), run(emu, wl)
    assert bo.trace(0)["status"] == K.STATUS_ENDED
    compare(bo, be, wl.n_instances)
    ns = be.net_state(0)
    assert len(ns["events"]) == 0 and len(ns["l1_messages"]) == 0
    # final storage == the populated snapshot again
    fs = {(bytes(s["address"]), s["key"].tobytes()): s["value"].tobytes() for s in ns["final_storage"]}
    for s in wl.storage[0]:
        assert fs[(bytes(s["address"]), s["key"].tobytes())] == s["value"].tobytes()


def test_partial_runs_net_open_frames_as_kept(oracle, emu, isa):
    """Stopping inside the inner frame: the reference's flatten would assert (open frames); both sides net as if the
    open frames were kept."""
    wl = synth.nested_frames(isa, outer=K.RET_PANIC, inner=K.RET_OK)
    for cycles in (1, 5, 22, 24, 27):
        bo, be = run(oracle, wl, cycles), run(emu, wl, cycles)
        compare(bo, be, wl.n_instances)


def test_l2_block_net_states(oracle, emu, isa):
    wl = synth.make(4, isa, n_instances=5)
    bo, be = run(oracle, wl), run(emu, wl)
    compare(bo, be, wl.n_instances)
    ns = be.net_state(0)
    assert len(ns["storage_history"]) > 50 and len(ns["events"]) > 10 and ((ns["storage_history"]["bools"] & K.LQ_ROLLBACK) != 0).any()


def test_failed_instance_has_no_net_state(oracle, emu, isa):
    wl = synth.make(2, isa, n_instances=2)
    wl.preimages = wl.preimages[:1]  # the second far call's code hash becomes unknown -> ZKW_STATUS_UNKNOWN_CODE_HASH
    bo, be = run(oracle, wl), run(emu, wl)
    assert be.trace(0)["status"] == K.STATUS_UNKNOWN_CODE_HASH
    for b in (bo, be):
        with pytest.raises(K.ZkwError):
            b.net_state(0)


def test_net_state_capacity_overflow_is_reported(emu, isa):
    wl = synth.make(4, isa, n_instances=2)
    wl.limits["max_callstack_depth"] = 1  # frame marks: depth + 2 = 3, enough here; shrink the aux index list instead
    wl.limits["max_aux_events"] = 4
    b = emu.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    with pytest.raises(K.ZkwError):
        b.net_state(0)


def test_golden_net_state_digests(oracle, emu, isa):
    """Regression pins (tests/golden/net_state_digests.json, generated by tests/golden/make_golden.py from the oracle):
    the oracle still produces them, and so does the product."""
    import json
    from golden.make_golden import net_state_cases, net_state_digest
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "net_state_digests.json")))
    cases = net_state_cases(isa)
    assert sorted(cases) == sorted(golden)
    for name, wl in cases.items():
        assert net_state_digest(oracle, wl) == golden[name], name
        assert net_state_digest(emu, wl) == golden[name], name
