"""The C++ host mirror of the reference surface (era-zk_evm_amd/host/zk_evm.hpp): `BatchedVmState::cycle()`
replays a finished run into a `VmWitnessTracer` + `EventSink`.  Its stream of outward calls — every
argument of every callback, including the fully rebuilt VmLocalState at each start/end_execution_cycle
— must equal, call for call, what the oracle's restated `cycle()` calls directly."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from era_zk_evm_amd import capi as K, synth

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


def build_replay_lib():
    src = os.path.join(HERE, "host", "replay_lib.cpp")
    out = os.path.join(HERE, "host", "libzkw_host_replay.so")
    deps = [src, os.path.join(HERE, "..", "era-zk_evm_amd", "host", "zk_evm.hpp"), os.path.join(HERE, "..", "oracle", "callback_log.hpp"),
            os.path.join(HERE, "..", "include", "zkw.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", out, src], check=True)
    return C.CDLL(out)


def oracle_callback_log(oracle, wl):
    b = oracle.create_batch(wl)
    oracle.lib.zkwo_batch_enable_callback_log(b.h, C.c_int(1))
    b.reset()
    b.run(wl.n_cycles)
    logs = []
    for i in range(wl.n_instances):
        p, n = C.POINTER(C.c_uint64)(), C.c_uint32()
        assert oracle.lib.zkwo_batch_get_callback_log(b.h, C.c_uint32(i), C.byref(p), C.byref(n)) == 0
        logs.append(np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.uint64))
    return b, logs


def replay_log(lib, backend_batch, wl, i, trace=None):
    """the host mirror's callback log for instance i: replayed from `trace` (a filled InstanceTraceC, e.g. one rebuilt from the
    delivery ring) or from zkw_batch_get_instance_trace of `backend_batch`"""
    t = trace
    if t is None:
        t = K.InstanceTraceC()
        backend_batch.be.call("batch_get_instance_trace", backend_batch.h, C.c_uint32(i), C.byref(t))
    words = np.concatenate([np.zeros((0, 4), dtype="<u8")] + [np.ascontiguousarray(b, dtype="<u8").reshape(-1, 4) for b in wl.blobs]) if wl.blobs else np.zeros((1, 4), "<u8")
    first = np.zeros(len(wl.blobs) + 1, dtype=np.uint32)
    length = np.zeros(len(wl.blobs) + 1, dtype=np.uint32)
    off = 0
    for j, b in enumerate(wl.blobs):  # blob ids start at 1 (0 = the zero page)
        first[j + 1], length[j + 1] = off, len(b)
        off += len(b)
    words = np.ascontiguousarray(words)
    cap = 8 * (t.n_cycles + 8) + 4 * (t.n_mem + t.n_log + t.n_aux)
    out = np.zeros(cap, dtype=np.uint64)
    n, rc = C.c_uint32(), C.c_int()
    st = np.ascontiguousarray(wl.states[i:i + 1])
    inner = np.ascontiguousarray(wl.inner[i])
    r = lib.zkw_host_replay_callback_log(K._ptr(st), K._ptr(inner), C.byref(t), K._ptr(words), K._ptr(first), K._ptr(length), C.c_uint32(len(first)),
                                         K._ptr(out), C.c_uint32(cap), C.byref(n), C.byref(rc))
    assert r == 0 and n.value <= cap
    return out[:n.value], rc.value


@pytest.fixture(scope="module")
def replay_lib():
    return build_replay_lib()


@pytest.fixture(scope="module")
def emu(isa):
    import build_emu
    be = K.Backend(build_emu.build(), "zkw_").open(isa)
    yield be
    be.close()


CASES = {
    "cfg1": lambda isa: synth.make(1, isa, n_instances=4),
    "cfg2": lambda isa: synth.make(2, isa, n_instances=4),
    "cfg3": lambda isa: synth.make(3, isa, n_instances=2, keccak_k=(1, 2, 3, 1), sha_rounds=(1, 2, 3, 5)),
    "cfg4": lambda isa: synth.make(4, isa, n_instances=3, n_cycles=1024),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_replay_of_oracle_trace_reproduces_its_own_calls(oracle, replay_lib, isa, name):
    """mirror(oracle trace) == oracle's direct calls: pins the record format + replay logic."""
    wl = CASES[name](isa)
    b, logs = oracle_callback_log(oracle, wl)
    for i in range(wl.n_instances):
        got, rc = replay_log(replay_lib, b, wl, i)
        assert len(got) == len(logs[i]) and (got == logs[i]).all(), "%s instance %d: first difference at call %d" % (
            name, i, int(np.flatnonzero(got[:min(len(got), len(logs[i]))] != logs[i][:min(len(got), len(logs[i]))])[0]) if len(got) and len(logs[i]) else -1)


@pytest.mark.parametrize("name", ["cfg2", "cfg4"])
def test_replay_of_kernel_trace_equals_reference_calls(oracle, emu, replay_lib, isa, name):
    """mirror(kernel trace) == oracle's direct calls (kernel sources compiled for the CPU stand-in here;
    tests/test_gpu_parity.py::test_host_replay_on_gpu does the same with the real GPU run)."""
    wl = CASES[name](isa)
    _, logs = oracle_callback_log(oracle, wl)
    be = emu.create_batch(wl)
    be.reset(); be.run(wl.n_cycles); be.sync()
    for i in range(wl.n_instances):
        got, rc = replay_log(replay_lib, be, wl, i)
        assert len(got) == len(logs[i]) and (got == logs[i]).all()


def check_tracer_calls_from_the_ring(oracle, prod, replay_lib, isa, names, host_threads=2, lanes=None):
    """The drop-in boundary end to end: steps run on the device, DELIVERED into the pinned ring (zkw_delivery_submit), every
    instance's trace rebuilt from the ring (zkw_delivery_get_instance_trace) and replayed through the host mirror of
    VmState::cycle — the ten VmWitnessTracer callbacks, argument for argument — against the calls the oracle's own cycle() made."""
    wls = [CASES[n](isa) for n in names]
    logs = [oracle_callback_log(oracle, CASES[n](isa))[1] for n in names]
    bps = []
    for wl in wls:
        if lanes is not None:
            wl.limits["lanes_per_wave"] = lanes
        b = prod.create_batch(wl)
        b.reset(); b.run(wl.n_cycles)
        bps.append(b)
    dv = K.Delivery(prod, 2, K.Delivery.worst_case_bytes(prod, bps), host_threads)
    t = dv.submit(bps)
    assert dv.wait(t)["overflow"] == 0
    for bi, (wl, lg) in enumerate(zip(wls, logs)):
        for i in range(wl.n_instances):
            tr = K.InstanceTraceC()
            prod.call("delivery_get_instance_trace", dv.h, C.c_uint32(t), C.c_uint32(bi), C.c_uint32(i), C.byref(tr))
            got, rc = replay_log(replay_lib, None, wl, i, trace=tr)
            assert len(got) == len(lg[i]) and (got == lg[i]).all(), (names[bi], i)
    dv.release(t)
    dv.close()
    for b in bps:
        b.destroy()


def test_tracer_calls_replayed_from_the_delivery_ring_equal_reference_calls(oracle, emu, replay_lib, isa):
    check_tracer_calls_from_the_ring(oracle, emu, replay_lib, isa, ["cfg2", "cfg4", "cfg3"])


def test_replay_surfaces_reference_errors(oracle, replay_lib, isa):
    wl = synth.make(2, isa, n_instances=1)
    wl.preimages = wl.preimages[1:]
    b, logs = oracle_callback_log(oracle, wl)
    got, rc = replay_log(replay_lib, b, wl, 0)
    assert rc == K.STATUS_UNKNOWN_CODE_HASH  # BatchedVmState::cycle() reports the reference's Err at the failing cycle
    assert len(got) == len(logs[0]) and (got == logs[0]).all()


@pytest.mark.parametrize("make", [lambda isa: synth.make(4, isa, n_instances=3), lambda isa: synth.nested_frames(isa, outer=K.RET_PANIC, inner=K.RET_OK),
                                  lambda isa: synth.nested_frames(isa, outer=K.RET_OK, inner=K.RET_REVERT)])
def test_host_event_sink_mirror_flattens_to_the_device_net_state(oracle, emu, replay_lib, isa, make):
    """BatchedVmState's replay into the host mirror of InMemoryEventSink ends where the reference's sink would:
    flatten() == the events part of zkw_batch_get_net_state (computed on the device / by the oracle)."""
    wl = make(isa)
    be = emu.create_batch(wl)
    be.reset(); be.run(wl.n_cycles); be.sync()
    bo = oracle.create_batch(wl)
    bo.reset(); bo.run(wl.n_cycles); bo.sync()
    for i in range(wl.n_instances):
        t = K.InstanceTraceC()
        be.be.call("batch_get_instance_trace", be.h, C.c_uint32(i), C.byref(t))
        cap = 2 * t.n_log + 8
        hist = np.zeros(cap, dtype=K.LOG_QUERY)
        ev = np.zeros(cap, dtype=K.EVENT_MESSAGE)
        l1 = np.zeros(cap, dtype=K.EVENT_MESSAGE)
        nh, ne, nl = C.c_uint32(), C.c_uint32(), C.c_uint32()
        st = np.ascontiguousarray(wl.states[i:i + 1])
        inner = np.ascontiguousarray(wl.inner[i])
        rc = replay_lib.zkw_host_replay_event_sink(K._ptr(st), K._ptr(inner), C.byref(t), C.c_uint32(int(wl.states[i]["callstack_depth"])),
                                                   C.c_uint8(int(isa.table["consts"]["event_aux_byte"][0])), K._ptr(hist), C.c_uint32(cap), C.byref(nh),
                                                   K._ptr(ev), C.c_uint32(cap), C.byref(ne), K._ptr(l1), C.c_uint32(cap), C.byref(nl))
        assert rc == 0
        for ns in (be.net_state(i), bo.net_state(i)):
            assert ns["event_history"].tobytes() == hist[: nh.value].tobytes()
            assert ns["events"].tobytes() == ev[: ne.value].tobytes()
            assert ns["l1_messages"].tobytes() == l1[: nl.value].tobytes()


def test_mirror_dumps_pages_through_the_c_abi(oracle, isa):
    """`vm.memory.dump_page_content(page, range)` of the mirror (zk_evm::SimpleMemory over the library's get_page entry)
    = the big-endian bytes of the words the C ABI returns (memory.rs:300-314)."""
    lib = build_replay_lib()
    wl = synth.make(2, isa, n_instances=3)
    b = oracle.create_batch(wl)
    b.reset()
    b.run(wl.n_cycles)
    fn = C.cast(getattr(oracle.lib, oracle.prefix + "batch_get_page"), C.c_void_p)
    page = synth.BOOTLOADER_BASE_PAGE + 2  # the bootloader's heap
    out = np.zeros((40, 32), dtype=np.uint8)
    assert lib.zkw_host_dump_page(fn, b.h, C.c_uint32(1), C.c_uint32(page), C.c_uint32(3), C.c_uint32(43), K._ptr(out)) == 0
    words = b.page(1, page, 3, 40)
    assert words.any()
    for k in range(40):
        assert int.from_bytes(out[k].tobytes(), "big") == K.u256_to_int(words[k])
