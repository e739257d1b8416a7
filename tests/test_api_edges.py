"""Edge cases of the C-ABI (product sources in the emulation build): argument errors are error codes, capacity
overruns are per-instance ZKW_STATUS_LIMIT — never a crash, never silent truncation."""
import ctypes as C
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


def test_zero_instances_is_rejected(emu):
    lim = np.zeros(1, dtype=K.LIMITS)
    lim["max_cycles"] = 8
    h = C.c_void_p()
    rc = emu.fn("batch_create")(emu.ctx, C.c_uint32(0), K._ptr(lim), C.byref(h))
    assert rc == K.ERR_INVALID and not h.value


def test_calls_in_the_wrong_order_are_error_codes(emu, isa):
    wl = synth.make(1, isa, n_instances=2)
    b = emu.create_batch(wl)  # uploaded and reset, not run
    t = K.InstanceTraceC()
    assert emu.fn("batch_get_instance_trace")(b.h, C.c_uint32(0), C.byref(t)) == K.ERR_NOT_RUN
    st = np.zeros(1, dtype=K.RUN_STATS)
    assert emu.fn("batch_get_stats")(b.h, K._ptr(st)) == K.ERR_NOT_RUN
    assert emu.fn("batch_commit")(b.h, C.c_uint32(7), C.c_void_p(None)) == K.ERR_NOT_RUN
    assert emu.fn("batch_run")(b.h, C.c_uint32(0), C.c_void_p(None)) == K.ERR_LIMIT                       # zero cycles
    assert emu.fn("batch_run")(b.h, C.c_uint32(wl.limits["max_cycles"] + 1), C.c_void_p(None)) == K.ERR_LIMIT
    b.run(wl.n_cycles)
    b.sync()
    assert emu.fn("batch_get_instance_trace")(b.h, C.c_uint32(2), C.byref(t)) == K.ERR_INVALID           # instance out of range
    assert emu.fn("batch_run")(b.h, C.c_uint32(1), C.c_void_p(None)) == K.ERR_LIMIT                       # record capacity used up
    assert emu.fn("batch_get_instance_trace")(C.c_void_p(None), C.c_uint32(0), C.byref(t)) == K.ERR_INVALID


def test_single_instance_and_ragged_waves(oracle, emu, isa):
    for n in (1, 3):
        wl = synth.make(2, isa, n_instances=n)
        bo = oracle.create_batch(wl); bo.reset(); bo.run(wl.n_cycles); bo.sync()
        be = emu.create_batch(wl); be.reset(); be.run(wl.n_cycles); be.sync()
        for i in range(n):
            ok, why = K.traces_equal(bo.trace(i), be.trace(i))
            assert ok, (n, i, why)


@pytest.mark.parametrize("field,value", [("max_mem_queries", 8), ("max_log_queries", 2), ("max_aux_events", 2), ("storage_journal", 1),
                                         ("stack_words", 2), ("max_far_frames", 1), ("max_callstack_depth", 1)])
def test_capacity_overruns_are_limit_statuses(oracle, emu, isa, field, value):
    """every zkw_limits field: too small a value ends the affected instances with ZKW_STATUS_LIMIT; the cycles they did
    complete are still bit-exact"""
    wl = synth.make(4, isa, n_instances=3)
    ref = synth.make(4, isa, n_instances=3)
    wl.limits[field] = value
    be = emu.create_batch(wl); be.reset(); be.run(wl.n_cycles); be.sync()
    bo = oracle.create_batch(ref); bo.reset(); bo.run(ref.n_cycles); bo.sync()
    hit = 0
    for i in range(3):
        te, to = be.trace(i), bo.trace(i)
        assert te["status"] in (K.STATUS_LIMIT, to["status"])
        if te["status"] == K.STATUS_LIMIT:
            hit += 1
            n = te["n_cycles"]
            assert n < to["n_cycles"]
            assert te["records"].tobytes() == to["records"][:n].tobytes()
    assert hit > 0, "the workload never reached this limit: pick a smaller value"


def test_storage_table_too_small_for_the_snapshot_is_an_argument_error(emu, isa):
    wl = synth.make(4, isa, n_instances=2)
    wl.limits["storage_slots"] = 256  # the snapshot alone holds 258 slots
    with pytest.raises(K.ZkwError):
        emu.create_batch(wl)


def test_heap_image_longer_than_the_page_is_an_argument_error(emu, isa):
    wl = synth.make(4, isa, n_instances=2)
    wl.limits["heap_words"] = 200  # the image holds 256 words
    with pytest.raises(K.ZkwError):
        emu.create_batch(wl)


def test_download_all_reports_the_trace_volume(emu, isa):
    wl = synth.make(2, isa, n_instances=4)
    b = emu.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    nb, ms = C.c_uint64(0), C.c_double(0)
    emu.call("batch_download_all", b.h, C.byref(nb), C.byref(ms))
    st = b.stats()
    # tails 32 B per cycle + 32 B per register delta + the query records (+ the directory)
    expect = 32 * int(st["cycles"]) + 32 * int(st["reg_deltas"]) + 48 * int(st["mem_queries"]) + 128 * int(st["log_queries"]) + 256 * int(st["aux_events"])
    assert expect <= nb.value <= expect + 4 * (wl.n_cycles + 1) * 16


def test_trace_pointers_stay_valid_across_more_than_64_waves(emu, isa):
    """include/zkw.h: the arrays of a zkw_instance_trace stay valid until the next run, reset or destroy.  With one lane per
    wave every instance is its own wave: fetch the traces of 80 waves keeping the RAW pointers (as the C++ mirror does),
    then re-read the first ones through those pointers — the runtime must not have evicted them (round-1 finding)."""
    import ctypes as C
    wl = synth.make(2, isa, n_instances=80)
    wl.limits["lanes_per_wave"] = 1
    b = emu.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    raw, copies = [], []
    for i in range(wl.n_instances):
        t = K.InstanceTraceC()
        emu.call("batch_get_instance_trace", b.h, C.c_uint32(i), C.byref(t))
        raw.append(t)
        copies.append((C.string_at(t.records, t.n_cycles * K.CYCLE_RECORD.itemsize), C.string_at(t.mem, t.n_mem * K.MEM_QUERY.itemsize)))
    for i in (0, 1, 17, 63, 64, 79):
        t = raw[i]
        assert C.string_at(t.records, t.n_cycles * K.CYCLE_RECORD.itemsize) == copies[i][0], i
        assert C.string_at(t.mem, t.n_mem * K.MEM_QUERY.itemsize) == copies[i][1], i
    b.destroy()


def test_malformed_isa_tables_are_rejected(isa):
    """zkw_ctx_set_isa refuses a table whose constants the kernel would evaluate blindly: a condition_lut row that is none of
    the eight Condition truth tables (cycle.rs:193-209), forwarding codes that collide (far_call.rs:255, ret.rs:59), a zero
    MAX_OFFSET_FOR_ADD_SUB (ptr.rs:47) — the round-4 advisor finding."""
    import build_emu
    be = K.Backend(build_emu.build(), "zkw_")
    be.call("ctx_create", C.c_int(0), C.byref(be.ctx))

    def rejected(edit, needle):
        t = isa.table.copy()
        edit(t["consts"][0])
        rc = be.fn("ctx_set_isa")(be.ctx, K._ptr(t))
        msg = (be.fn("last_error")(be.ctx) or b"").decode()
        assert rc == K.ERR_INVALID and needle in msg, (rc, msg)

    def bad_lut(c):
        c["condition_lut"] = (int(c["condition_lut"]) & ~(0xFF << 16)) | (0x5A << 16)  # row 2: no Condition
    rejected(bad_lut, "names no Condition")

    def same_codes(c):
        c["forwarding_codes"] = 0 | (1 << 8) | (1 << 16)
    rejected(same_codes, "distinct")

    def zero_max(c):
        c["max_offset_for_add_sub"] = 0
    rejected(zero_max, "max_offset_for_add_sub")
    be.call("ctx_set_isa", be.ctx, K._ptr(isa.table))  # the default table still passes
    for seed in (1, 2):  # ... and so do the tables of the metamorphic tests
        v = K.Isa.variant_of_default(seed, swap_forwarding=True, permute_conditions=True, shift_registers=True)
        be.call("ctx_set_isa", be.ctx, K._ptr(v.table))
    be.close()
