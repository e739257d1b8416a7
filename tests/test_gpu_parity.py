"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle on the
same seeded inputs — bit-exact over every record of every instance."""
import numpy as np
import pytest

from era_zk_evm_amd import capi as K, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def product(isa):
    be = K.load_product().open(isa)
    yield be
    be.close()


def _run(backend, wl, lanes=0):
    wl.limits["lanes_per_wave"] = lanes
    b = backend.create_batch(wl)
    b.reset()
    b.run(wl.n_cycles)
    b.sync()
    return b


def _compare(oracle, product, wl, lanes=0, sample=None, commitments=False):
    bo = _run(oracle, wl)
    bp = _run(product, wl, lanes)
    idx = range(wl.n_instances) if sample is None else sample
    for i in idx:
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, "%s instance %d (lanes=%d): %s" % (wl.name, i, lanes, why)
    assert int(bp.stats()["cycles"]) == int(bo.stats()["cycles"])
    if commitments:  # the three queue digests of EVERY instance (the oracle run is there anyway)
        assert np.array_equal(bo.commitments(), bp.commitments()), wl.name
    bo.destroy()
    bp.destroy()


def test_cfg0_nop_add(oracle, product, isa):
    _compare(oracle, product, synth.make(0, isa))


@pytest.mark.parametrize("lanes", [0, 1, 16, 64])
def test_cfg1_arith_256x256(oracle, product, isa, lanes):
    _compare(oracle, product, synth.make(1, isa), lanes)


@pytest.mark.parametrize("lanes", [0, 8, 64])
def test_cfg2_mixed(oracle, product, isa, lanes):
    _compare(oracle, product, synth.make(2, isa, n_instances=320), lanes)


def test_cfg2_ragged_last_wave(oracle, product, isa):
    # instance count that is not a multiple of the lanes per wave
    _compare(oracle, product, synth.make(2, isa, n_instances=77), 64)


def test_cfg2_full_size_sampled(oracle, product, isa):
    # BASELINE config: 4096 x 256 = 1M cycles; every 37th instance compared record by record
    wl = synth.make(2, isa, n_instances=4096)
    _compare(oracle, product, wl, 0, sample=range(0, 4096, 37), commitments=True)


def test_rerun_after_reset_is_identical(product, isa):
    wl = synth.make(2, isa, n_instances=64)
    b = product.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    t1 = [b.trace(i) for i in (0, 17, 63)]
    b.reset(); b.run(wl.n_cycles); b.sync()
    t2 = [b.trace(i) for i in (0, 17, 63)]
    for a, c in zip(t1, t2):
        ok, why = K.traces_equal(a, c)
        assert ok, why


@pytest.mark.parametrize("cfg,kw", [(2, dict(n_instances=192)), (4, dict(n_instances=96, n_cycles=1024))])
def test_reset_restores_everything_a_run_changed(oracle, product, isa, cfg, kw):
    """reset -> run three times on one batch (the middle run is partial, so it dirties other words / slots): a reset
    restores only what the run marked and the first launch after it starts from the pristine images — the full runs
    must reproduce the oracle's traces and commitments (cfg 4: storage writes, rollbacks, events)."""
    wl = synth.make(cfg, isa, **kw)
    bo = oracle.create_batch(wl)
    bo.reset(); bo.run(wl.n_cycles)
    bp = product.create_batch(wl)
    for rnd in range(3):
        bp.reset(); bp.run(wl.n_cycles if rnd != 1 else wl.n_cycles // 2); bp.sync()
        if rnd == 1:
            continue
        for i in range(0, wl.n_instances, 7):
            ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
            assert ok, "cfg %d round %d instance %d: %s" % (cfg, rnd, i, why)
        assert np.array_equal(bo.commitments(), bp.commitments()), (cfg, rnd)


def test_split_run_equals_single_run(product, isa):
    wl = synth.make(2, isa, n_instances=64)
    b = product.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    whole = [b.trace(i) for i in (0, 31)]
    b.reset(); b.run(100); b.run(wl.n_cycles - 100); b.sync()
    parts = [b.trace(i) for i in (0, 31)]
    for a, c in zip(whole, parts):
        ok, why = K.traces_equal(a, c)
        assert ok, why


@pytest.mark.parametrize("lanes", [0, 64])
def test_cfg3_precompiles(oracle, product, isa, lanes):
    _compare(oracle, product, synth.make(3, isa, n_instances=96, keccak_k=(1, 2, 8, 3), sha_rounds=(1, 2, 8, 5)), lanes)


def test_cfg3_baseline_sizes_sampled(oracle, product, isa):
    # BASELINE sizes: keccak 136*{1,8,64,512} bytes, sha256 {1,8,64,157} rounds; 128 instances on the GPU,
    # a sample of them against the oracle, and every digest against hashlib / an independent Keccak
    import hashlib
    from test_oracle_precompiles import _keccak256_py
    wl = synth.make(3, isa, n_instances=512)  # one GPU's shard of BASELINE configs[3] (4096 instances over 8 GPUs)
    bp = _run(product, wl)
    bo = _run(oracle, wl)
    for i in (0, 63, 64, 127, 255, 256, 448, 511):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, "instance %d: %s" % (i, why)
    for i in range(0, 512, 37):
        t = bp.trace(i)
        writes = [q for q in t["mem"] if (q["meta"] >> K.MQ_KIND_SHIFT) == 2]
        data = wl.heap_bytes[i].tobytes()
        for (start, length, _), q in zip(wl.sha_messages, writes[:4]):
            assert K.u256_to_int(q["value"]).to_bytes(32, "big") == hashlib.sha256(data[start:start + length]).digest()
        for (start, length, _), q in zip(wl.keccak_messages[:3], writes[4:7]):
            assert K.u256_to_int(q["value"]).to_bytes(32, "big") == _keccak256_py(data[start:start + length])


@pytest.mark.parametrize("lanes", [0, 16, 64])
def test_cfg4_l2_block(oracle, product, isa, lanes):
    _compare(oracle, product, synth.make(4, isa, n_instances=192, n_cycles=1024), lanes)


def test_cfg4_full_size_sampled(oracle, product, isa):
    wl = synth.make(4, isa, n_instances=4096, n_cycles=1024)
    _compare(oracle, product, wl, 0, sample=range(5, 4096, 211), commitments=True)


def test_divergent_tapes_in_one_wave(oracle, product, isa):
    """Instances of one wave running DIFFERENT programs (lanes diverge on every opcode): half the lanes run
    the cfg-4 tape, the others a cfg-1-style arithmetic tape through a second code page."""
    wl = synth.make(4, isa, n_instances=64, n_cycles=512)
    other = synth.arith_tape(isa, 512, synth.ScalarRng(99))
    wl.blobs.append(K.pack_code(other))
    alt = len(wl.blobs) - 1
    for i in range(1, 64, 2):
        wl.code_pages.append((i, 1, synth.BOOTLOADER_CODE_PAGE, alt))
    _compare(oracle, product, wl, 64)


def test_status_codes(oracle, product, isa):
    wl = synth.make(2, isa, n_instances=64)
    wl.preimages = wl.preimages[1:]
    bo, bp = _run(oracle, wl), _run(product, wl, 64)
    for i in (0, 31, 63):
        to, tp = bo.trace(i), bp.trace(i)
        assert tp["status"] == K.STATUS_UNKNOWN_CODE_HASH
        ok, why = K.traces_equal(to, tp)
        assert ok, why
    wl = synth.make(0, isa, n_cycles=64)
    wl.blobs[0] = K.pack_code([isa.enc(K.OP_ADD, src0=1, src1=2, dst0=3)] * 5 + [isa.enc(K.OP_RET, variant=K.RET_OK, src0=0)])
    to, tp = _run(oracle, wl).trace(0), _run(product, wl).trace(0)
    assert tp["status"] == K.STATUS_ENDED and tp["n_cycles"] == 6
    ok, why = K.traces_equal(to, tp)
    assert ok, why


@pytest.mark.parametrize("cfg,kw", [(2, dict(n_instances=200)), (4, dict(n_instances=96, n_cycles=512)),
                                    (3, dict(n_instances=64, keccak_k=(1, 2, 3, 1), sha_rounds=(1, 2, 3, 5)))])
def test_queue_commitments(oracle, product, isa, cfg, kw):
    wl = synth.make(cfg, isa, **kw)
    bo, bp = _run(oracle, wl), _run(product, wl)
    assert np.array_equal(bo.commitments(), bp.commitments())


def test_host_replay_on_gpu(oracle, product, isa):
    """End to end: GPU run -> C ABI trace -> C++ host mirror (VmWitnessTracer/EventSink replay) == the calls the
    oracle's restated cycle() makes directly."""
    from test_host_replay import build_replay_lib, oracle_callback_log, replay_log
    lib = build_replay_lib()
    wl = synth.make(4, isa, n_instances=70, n_cycles=1024)
    _, logs = oracle_callback_log(oracle, wl)
    bp = _run(product, wl, 64)
    for i in (0, 1, 63, 64, 69):
        got, rc = replay_log(lib, bp, wl, i)
        assert len(got) == len(logs[i]) and (got == logs[i]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [64, 0])
def test_tracer_calls_replayed_from_the_delivery_ring_on_gpu(oracle, product, isa, lanes):
    """The north star's boundary end to end on the device: steps of three different workloads delivered into the pinned ring by the
    pack kernel, every instance's trace rebuilt from the ring and replayed through the host mirror of VmState::cycle: the ten
    VmWitnessTracer callbacks, argument for argument, equal the calls the oracle's own cycle() made."""
    from test_host_replay import build_replay_lib, check_tracer_calls_from_the_ring
    check_tracer_calls_from_the_ring(oracle, product, build_replay_lib(), isa, ["cfg2", "cfg4", "cfg3"], host_threads=4, lanes=lanes)


@pytest.mark.parametrize("cfg,kw", [(1, dict()), (2, dict(n_instances=320)), (4, dict(n_instances=128, n_cycles=512))])
def test_generic_per_lane_path_forced(oracle, product, isa, cfg, kw):
    """ZKW_OPT_DEBUG_FLAGS = 4 disables the wave-uniform fast path: the fully per-lane decode must give the same bits."""
    product.set_option(K.OPT_DEBUG_FLAGS, 4)
    try:
        _compare(oracle, product, synth.make(cfg, isa, **kw), 64)
    finally:
        product.set_option(K.OPT_DEBUG_FLAGS, 0)


@pytest.mark.parametrize("cfg,kw", [(1, dict()), (2, dict(n_instances=320)), (3, dict(n_instances=64, keccak_k=(1, 2, 8, 3), sha_rounds=(1, 2, 8, 5))),
                                    (4, dict(n_instances=128, n_cycles=512))])
def test_variant_group_path_forced(oracle, product, isa, cfg, kw):
    """ZKW_OPT_DEBUG_FLAGS bit 24 sends every group of a light opcode through the variant-group path (per-lane operand decode,
    waterfall register access in zkw_vec_exec) even on a shared tape: the same bits as the scalar decode."""
    product.set_option(K.OPT_DEBUG_FLAGS, 1 << 24)
    try:
        _compare(oracle, product, synth.make(cfg, isa, **kw), 64)
    finally:
        product.set_option(K.OPT_DEBUG_FLAGS, 0)


@pytest.mark.parametrize("seed", [0xF101, 0xF102])
def test_fuzz_tapes_variant_groups_forced(oracle, product, isa, seed):
    """The fuzz tapes (every opcode variant, operand mode and failure path) with every light group on the variant path."""
    wl = synth.fuzz_workload(isa, n_instances=256, n_ops=96, seed=seed)
    bo = _run(oracle, wl)
    product.set_option(K.OPT_DEBUG_FLAGS, 1 << 24)
    try:
        bp = _run(product, wl, 64)
    finally:
        product.set_option(K.OPT_DEBUG_FLAGS, 0)
    compared = 0
    for i in range(wl.n_instances):
        tp = bp.trace(i)
        if int(tp["status"]) == K.STATUS_LIMIT:
            continue
        ok, why = K.traces_equal(bo.trace(i), tp)
        assert ok, "fuzz %x instance %d: %s" % (seed, i, why)
        compared += 1
    assert compared * 8 > wl.n_instances * 7
    bo.destroy()
    bp.destroy()


def test_graph_replayed_step(oracle, product, isa):
    """zkw_batch_step: the first call runs eagerly and captures a hipGraph, later calls replay it — same bits."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    wl = synth.make(2, isa, n_instances=256)
    bo = _run(oracle, wl)
    bp = product.create_batch(wl)
    for _ in range(4):
        product.call("batch_step", bp.h, C.c_uint32(wl.n_cycles), C.c_uint32(7), stream)
    bp.sync()
    for i in range(0, 256, 17):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, why
    cp = bp.commitments()
    assert np.array_equal(bo.commitments(), cp)
    assert float(bp.stats()["kernel_ms"]) > 0


def test_fused_step_of_several_batches(oracle, product, isa):
    """zkw_batches_step on the GPU: four batches of different shape and geometry in shared launches, bit-exact each."""
    wls = [synth.make(2, isa, n_instances=300), synth.make(4, isa, n_instances=64), synth.make(1, isa, n_instances=1000), synth.make(3, isa, n_instances=70)]
    cyc = max(w.n_cycles for w in wls)
    for w in wls:
        w.limits["max_cycles"] = cyc
    bos = []
    for w in wls:
        bo = oracle.create_batch(w)
        bo.reset(); bo.run(cyc); bo.sync()
        bos.append(bo)
    bps = [product.create_batch(w) for w in wls]
    for _ in range(3):
        product.step_many(bps, cyc, 7)
    for w, bo, bp in zip(wls, bos, bps):
        bp.sync()
        for i in range(0, w.n_instances, 7):
            ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
            assert ok, (w.name, i, why)
        assert np.array_equal(bo.commitments(), bp.commitments()), w.name


@pytest.mark.parametrize("lanes", [64, 16])
def test_every_record_of_whole_steps(oracle, product, isa, lanes):
    """Whole steps with room for helper waves (zkw_batches_step: the geometry of the driver's command), repeated so that the
    restore between them is part of it.  Every record of every instance against the oracle: divergent fuzz tapes (lanes
    that fail or end early beside the others), then cfg 2 (far calls, decommits chained by the helper wave)."""
    wl = synth.fuzz_workload(isa, n_instances=320, n_ops=96, seed=0xF0A1)
    bo = _run(oracle, wl)
    wl.limits["lanes_per_wave"] = lanes
    bp = product.create_batch(wl)
    for _ in range(2):
        product.step_many([bp], wl.n_cycles, 4)
    bp.sync()
    compared = 0
    for i in range(wl.n_instances):
        tp = bp.trace(i)
        if int(tp["status"]) == K.STATUS_LIMIT:
            continue
        ok, why = K.traces_equal(bo.trace(i), tp)
        assert ok, (i, why)
        compared += 1
    assert compared > 250
    bo.destroy(); bp.destroy()
    wl = synth.make(2, isa, n_instances=700)
    bo = _run(oracle, wl)
    wl.limits["lanes_per_wave"] = lanes
    bp = product.create_batch(wl)
    for _ in range(3):
        product.step_many([bp], wl.n_cycles, 7)
    bp.sync()
    for i in range(wl.n_instances):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, (i, why)
    assert np.array_equal(bo.commitments(), bp.commitments())
    bo.destroy(); bp.destroy()


def test_full_steps_after_partial_ones(oracle, product, isa):
    """zkw_batches_step: full steps after partial ones (which dirty other heap words and storage slots than a full run) and
    after plain reset + run calls must reproduce the oracle — the reset restores exactly what the previous run marked."""
    wls = [synth.make(2, isa, n_instances=200), synth.make(4, isa, n_instances=100)]
    cyc = max(w.n_cycles for w in wls)
    for w in wls:
        w.limits["max_cycles"] = cyc
    bos = []
    for w in wls:
        bo = oracle.create_batch(w)
        bo.reset(); bo.run(cyc); bo.sync()
        bos.append(bo)
    if True:
        bs = [product.create_batch(w) for w in wls]
        for cycles in (cyc, cyc // 3, cyc, 7, cyc):
            product.step_many(bs, cycles, 7)
            if cycles != cyc:
                continue
            for w, bo, b in zip(wls, bos, bs):
                b.sync()
                for i in range(0, w.n_instances, 3):
                    ok, why = K.traces_equal(bo.trace(i), b.trace(i))
                    assert ok, (w.name, i, why)
                assert np.array_equal(bo.commitments(), b.commitments()), w.name
        # a plain reset + run after fused steps, and a fused step after a plain run
        for b, w, bo in zip(bs, wls, bos):
            b.reset(); b.run(cyc // 2); b.sync()
        product.step_many(bs, cyc, 7)
        for w, bo, b in zip(wls, bos, bs):
            b.sync()
            for i in range(0, w.n_instances, 5):
                ok, why = K.traces_equal(bo.trace(i), b.trace(i))
                assert ok, (w.name, i, why)
            assert np.array_equal(bo.commitments(), b.commitments()), w.name


@pytest.mark.parametrize("seed,limits", [(0xF301, dict(max_far_frames=2)), (0xF302, dict(max_aux_events=6)), (0xF303, dict(max_reg_deltas=40, max_far_frames=3))])
def test_inline_decommit_chain_equals_the_post_run_chain_on_failing_instances(product, isa, seed, limits):
    """A far call chains its decommit into the running commitment inside the cycle kernel (zkw_batches_step); the same
    cycle can still fail afterwards — no arena slot, no aux / register-delta capacity left (ZKW_STATUS_LIMIT) — and a
    failed cycle leaves no records, so the chains computed from the streams after a run (zkw_batch_run + _commit) never
    see that decommit.  Both ways must agree for EVERY instance, the failed ones included."""
    wl = synth.fuzz_workload(isa, n_instances=512, n_ops=96, seed=seed)
    wl.limits.update(limits)
    b1, b2 = product.create_batch(wl), product.create_batch(wl)
    product.step_many([b1], wl.n_cycles, 7)  # inline
    b1.sync()
    b2.reset(); b2.run(wl.n_cycles); b2.sync()  # post-run chains (commitments() commits)
    c1, c2 = b1.commitments(), b2.commitments()
    failed = [i for i in range(wl.n_instances) if int(b1.trace(i)["status"]) == K.STATUS_LIMIT]
    assert failed, "the limits were meant to stop some instances"
    assert np.array_equal(c1, c2), [i for i in range(wl.n_instances) if not np.array_equal(c1[i], c2[i])][:8]
    b1.destroy(); b2.destroy()


@pytest.mark.parametrize("outer,inner,main_panics", [(K.RET_OK, K.RET_OK, False), (K.RET_PANIC, K.RET_OK, False), (K.RET_OK, K.RET_REVERT, False),
                                                     (K.RET_REVERT, K.RET_PANIC, False), (K.RET_OK, K.RET_OK, True)])
def test_net_states_nested_frames(oracle, product, isa, outer, inner, main_panics):
    """get_final_net_states on the device for nested near-call frames, 130 instances = 3 waves."""
    wl = synth.nested_frames(isa, outer=outer, inner=inner, main_panics=main_panics, n_instances=130)
    bo, bp = _run(oracle, wl), _run(product, wl)
    for i in range(0, 130, 3):
        ok, why = K.net_states_equal(bo.net_state(i), bp.net_state(i))
        assert ok, (i, why)


def test_net_states_l2_block(oracle, product, isa):
    wl = synth.make(4, isa, n_instances=256)
    bo, bp = _run(oracle, wl), _run(product, wl)
    for i in range(0, 256, 5):
        a, b = bo.net_state(i), bp.net_state(i)
        ok, why = K.net_states_equal(a, b)
        assert ok, (i, why)
    assert len(a["storage_history"]) > 50 and len(a["events"]) > 10


def test_net_states_partial_run(oracle, product, isa):
    wl = synth.nested_frames(isa, outer=K.RET_PANIC, inner=K.RET_OK, n_instances=70)
    for cycles in (5, 22, 27):
        bo = oracle.create_batch(wl); bo.reset(); bo.run(cycles); bo.sync()
        bp = product.create_batch(wl); bp.reset(); bp.run(cycles); bp.sync()
        for i in (0, 63, 64, 69):
            ok, why = K.net_states_equal(bo.net_state(i), bp.net_state(i))
            assert ok, (cycles, i, why)


def test_ecrecover_precompile(oracle, product, isa):
    """ecrecover through the VM on the GPU: 70 instances x 3 signatures (valid, out-of-range, off-curve), bit-exact vs
    the oracle and correct vs the independent Python implementation."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_ecrecover import make_signatures, check_against_python
    words, expect = make_signatures(70, 3, seed=0xEC7)
    wl = synth.ecrecover_workload(isa, words)
    bo, bp = _run(oracle, wl), _run(product, wl)
    for i in range(0, 70, 3):
        tp = bp.trace(i)
        ok, why = K.traces_equal(bo.trace(i), tp)
        assert ok, (i, why)
        check_against_python(wl, tp, expect[i])


def test_register_delta_capacity_is_a_limit_status(oracle, product, isa):
    """Delta-form CycleRecords: a wave that runs out of delta capacity fails all its lanes with ZKW_STATUS_LIMIT at the
    same cycle; the records before that cycle are intact (130 instances = 3 waves)."""
    wl = synth.make(1, isa, n_instances=130)
    wl.limits["max_reg_deltas"] = 10
    bp = _run(product, wl)
    bo = _run(oracle, synth.make(1, isa, n_instances=130))
    for i in (0, 1, 63, 64, 127, 128, 129):
        tp, to = bp.trace(i), bo.trace(i)
        assert tp["status"] == K.STATUS_LIMIT
        n = tp["n_cycles"]
        assert 0 < n < to["n_cycles"]
        assert tp["records"].tobytes() == to["records"][:n].tobytes()
    assert len({bp.trace(i)["n_cycles"] for i in range(0, 64)}) == 1  # wave-uniform


@pytest.mark.parametrize("field,value", [("max_mem_queries", 8), ("max_log_queries", 2), ("max_far_frames", 1), ("max_callstack_depth", 1), ("stack_words", 2)])
def test_capacity_overruns_are_limit_statuses(oracle, product, isa, field, value):
    """too small a zkw_limits value ends the affected instances with ZKW_STATUS_LIMIT (130 instances = 3 waves); the
    cycles they completed are bit-exact"""
    wl = synth.make(4, isa, n_instances=130)
    wl.limits[field] = value
    bp = _run(product, wl)
    bo = _run(oracle, synth.make(4, isa, n_instances=130))
    hit = 0
    for i in (0, 1, 63, 64, 100, 129):
        tp, to = bp.trace(i), bo.trace(i)
        assert tp["status"] in (K.STATUS_LIMIT, to["status"])
        if tp["status"] == K.STATUS_LIMIT:
            hit += 1
            n = tp["n_cycles"]
            assert n < to["n_cycles"]
            assert tp["records"].tobytes() == to["records"][:n].tobytes()
    assert hit > 0


@pytest.mark.parametrize("lens,unal", [((0, 1, 3, 50), (0, 1, 2, 3)), ((135, 136, 137, 200), (31, 5, 30, 7)), ((272, 7, 131, 408), (1, 2, 3, 4))])
def test_keccak_precompile_odd_lengths_and_alignments(oracle, product, isa, lens, unal):
    wl = synth.make(3, isa, n_instances=70, keccak_bytes=lens, keccak_unalign=unal, sha_rounds=(1, 1, 1, 1))
    bo, bp = _run(oracle, wl), _run(product, wl)
    for i in range(0, 70, 3):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, (i, why)


def test_commitments_of_an_instance_beyond_its_nominal_share(oracle, product, isa):
    """The streams of a wave are one pool (limits.max_*_queries x lanes): an instance may log more than its nominal share as
    long as the wave's stream holds it — its trace comes back in full, and so must its queue commitments (fuzz seed 0x400c
    of the round-3 campaign: instance 386 logs past max_log_queries next to lanes that hit the capacity; the per-instance
    index lists of the commitment pass once cut its log queue short).  Instances with ZKW_STATUS_LIMIT are excluded."""
    wl = synth.fuzz_workload(isa, n_instances=512, n_ops=96, seed=0x400C)
    bo = _run(oracle, wl)
    bp = _run(product, wl, 64)
    co, cp = bo.commitments(), bp.commitments()
    keep = np.array([int(bp.trace(i)["status"]) != K.STATUS_LIMIT for i in range(wl.n_instances)])
    per = int(bp.limits["max_log_queries"][0]) or None  # (the library writes the resolved limits back)
    beyond = [i for i in np.nonzero(keep)[0] if per is not None and len(bp.trace(int(i))["log"]) > per]
    assert keep.sum() > 400
    assert np.array_equal(co[keep], cp[keep]), np.argwhere((co != cp).any(axis=-1) & keep[:, None]).tolist()
    if per is not None:
        assert beyond, "the seed no longer drives an instance past its nominal share"
    bo.destroy(); bp.destroy()


@pytest.mark.parametrize("lanes", [2, 8, 5])
@pytest.mark.parametrize("lens,unal", [((0, 1, 3, 50), (0, 1, 2, 3)), ((135, 136, 137, 200), (31, 5, 30, 7)), ((272, 7, 131, 408), (1, 2, 3, 4)), ((1088, 271, 690, 32), (9, 31, 0, 16))])
def test_keccak_served_by_helper_waves(oracle, product, isa, lanes, lens, unal):
    """Batches of thin waves (<= 8 lanes): every cycle wave has a helper wave that runs its keccak256 calls 25 lanes per
    message (zkw_kh_helper) — reads witnessed by the helper at positions the requester allocated, digest back through LDS.
    Same traces as the oracle, run() and the whole step (helper also chains the decommits) alike, commitments included."""
    wl = synth.make(3, isa, n_instances=37, keccak_bytes=lens, keccak_unalign=unal, sha_rounds=(1, 2, 1, 1))
    bo = _run(oracle, wl)
    bp = _run(product, wl, lanes)
    for i in range(wl.n_instances):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, (i, why)
    assert np.array_equal(bo.commitments(), bp.commitments())
    bs = product.create_batch(wl)
    product.step_many([bs], wl.n_cycles, 7)
    bs.sync()
    for i in range(0, wl.n_instances, 5):
        ok, why = K.traces_equal(bo.trace(i), bs.trace(i))
        assert ok, (i, why)
    assert np.array_equal(bo.commitments(), bs.commitments())
    for b in (bo, bp, bs):
        b.destroy()


def test_thin_waves_without_room_for_helpers_keep_the_lane_path(oracle, product, isa):
    """More thin waves than a round of workgroups with helpers holds (2100 instances at 2 lanes per wave = 1050 cycle
    waves): the launch falls back to the ordinary geometry and every lane runs its own keccak256 — same traces."""
    wl = synth.make(3, isa, n_instances=2100, keccak_k=(1, 2, 1, 1), sha_rounds=(1, 1, 2, 1))
    bo = _run(oracle, wl)
    bp = _run(product, wl, 2)
    for i in list(range(0, 2100, 97)) + [2099]:
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, (i, why)
    assert int(bp.stats()["cycles"]) == int(bo.stats()["cycles"])
    bo.destroy(); bp.destroy()


@pytest.mark.parametrize("lanes", [2, 8])
def test_reference_keccak_kats_through_the_helper_waves(product, isa, lanes):
    """The reference's keccak256 cases (src/testing/tests/precompiles/keccak256.rs:144-196) through the lane-parallel
    sponge of the helper waves: digests against the literal SURVEY Appendix C values, no oracle in between."""
    from test_oracle_precompiles import KECCAK_KATS
    lens = (0, 50, 136, 200)
    for unalignment in (0, 31):
        wl = synth.make(3, isa, n_instances=19, keccak_bytes=lens, keccak_unalign=(unalignment,) * 4, sha_rounds=(1, 1, 1, 1))
        by = wl.heap_bytes
        for (start, length, _) in wl.keccak_messages:
            by[:, start - unalignment:start] = 0xFF
            by[:, start:start + length] = 123
        n, hw = by.shape[0], by.shape[1] // 32
        wl.heaps = np.ascontiguousarray(by.reshape(n, hw, 4, 8).view(">u8").reshape(n, hw, 4)[:, :, ::-1].astype("<u8"))
        bp = _run(product, wl, lanes)
        for i in range(wl.n_instances):
            t = bp.trace(i)
            writes = [q for q in t["mem"] if (q["meta"] >> K.MQ_KIND_SHIFT) == 2]
            for length, q in zip(lens, writes[4:8]):
                assert K.u256_to_int(q["value"]).to_bytes(32, "big").hex() == KECCAK_KATS[length], (i, length, unalignment)
        bp.destroy()


@pytest.mark.parametrize("seed,lanes,hook", [(0xF100, 64, 0), (0xF101, 64, 0), (0xF102, 0, 0), (0xF103, 16, 0), (0xF104, 64, 0), (0xF105, 64, 0), (0xF100, 64, 1 << 24)])
def test_uniform_fuzz_shared_tape(oracle, product, isa, seed, lanes, hook):
    """One random tape for every instance, per-instance registers and heaps (synth.uniform_fuzz): the lanes of a wave stay
    at one pc, so the cycle kernel's SHORT CYCLE executes what qualifies — ALU instructions with register / immediate /
    code-page operands, mul, shifts, heap and aux-heap accesses at per-lane offsets and alignments, conditional
    instructions that run in some lanes only, r0 destinations, dst0 == dst1 — and refuses, before it has written
    anything, what does not (a growing heap, an exception in one lane, stack operands).  Every record and query of every
    instance against the oracle; the last case runs the same tape with the short cycle switched off (test hook)."""
    wl = synth.uniform_fuzz(isa, n_instances=320, n_ops=192, seed=seed)
    bo = _run(oracle, wl)
    if hook:
        product.set_option(K.OPT_DEBUG_FLAGS, hook)
    try:
        bp = _run(product, wl, lanes)
    finally:
        if hook:
            product.set_option(K.OPT_DEBUG_FLAGS, 0)
    executed = 0
    for i in range(wl.n_instances):
        tp = bp.trace(i)
        assert int(tp["status"]) != K.STATUS_LIMIT
        ok, why = K.traces_equal(bo.trace(i), tp)
        assert ok, "uniform fuzz %x instance %d (lanes=%d): %s" % (seed, i, lanes, why)
        executed += len(tp["records"])
    assert executed > 150 * wl.n_instances
    assert np.array_equal(bo.commitments(), bp.commitments())
    bo.destroy(); bp.destroy()


@pytest.mark.parametrize("seed,lanes", [(0xF001, 64), (0xF002, 64), (0xF003, 16), (0xF004, 0)])
def test_fuzz_tapes(oracle, product, isa, seed, lanes):
    """Every instance runs its own tape of random valid encodings (synth.fuzz_workload): all 64 lanes of a wave diverge
    on every cycle, and the rarely used opcode variants / operand modes / failure paths are driven through the kernel
    and the oracle on the same inputs.  Instances that overran a capacity of the batch (ZKW_STATUS_LIMIT, a notion
    the reference does not have) are excluded; they must stay a small minority."""
    wl = synth.fuzz_workload(isa, n_instances=384, n_ops=96, seed=seed)
    bo = _run(oracle, wl)
    bp = _run(product, wl, lanes)
    limited = compared = executed = 0
    for i in range(wl.n_instances):
        tp = bp.trace(i)
        if int(tp["status"]) == K.STATUS_LIMIT:
            limited += 1
            continue
        ok, why = K.traces_equal(bo.trace(i), tp)
        assert ok, "fuzz %x instance %d (lanes=%d): %s" % (seed, i, lanes, why)
        compared += 1
        executed += len(tp["records"])
    assert limited * 8 < wl.n_instances, limited
    assert executed > 40 * compared  # the tapes survive: ~60 of 96 cycles on average
    bo.destroy()
    bp.destroy()


def test_cfg2_long_traces(oracle, product, isa):
    """The second shape of cfg 2: 256 instances x 4096 cycles, full waves — every record of every instance."""
    wl = synth.make(2, isa, n_instances=256, n_cycles=4096)
    _compare(oracle, product, wl, 64)


def test_reduce_commitments_rccl_world1(oracle, product, isa):
    """zkw_reduce_commitments (include/zkw.h) over an RCCL communicator of one rank: the same entry point bench.py
    drives at N GPUs — id exchange, size exchange, packed digests through ncclAllGather on the stream, counters
    through ncclAllReduce — against the oracle's digests and counters."""
    import torch
    wl = synth.make(2, isa, n_instances=96)
    bo = _run(oracle, wl)
    want = bo.commitments()  # [n][3][4]
    batches = [product.create_batch(wl) for _ in range(3)]
    stream = torch.cuda.Stream()
    product.step_many(batches, wl.n_cycles, 7, stream.cuda_stream)
    comm = K.Comm.rccl(product, 0, 1, K.Comm.unique_id(product))
    out = torch.zeros((1, 3, wl.n_instances, 2, 4), dtype=torch.int64, device="cuda")
    _, n_max, sizes, total = comm.reduce(batches, 0b110, gathered=out.data_ptr(), want_total=True, stream=stream.cuda_stream)
    stream.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    assert n_max == wl.n_instances and sizes == [wl.n_instances]
    for j in range(3):
        assert np.array_equal(got[0, j, :, 0], want[:, 1]) and np.array_equal(got[0, j, :, 1], want[:, 2])  # log, decommit
    so = bo.stats()
    assert int(total["cycles"]) == 3 * int(so["cycles"]) and int(total["mem_queries"]) == 3 * int(so["mem_queries"])
    assert int(total["instances_failed"]) == 0
    comm.close()


@pytest.mark.parametrize("unalignment", [0, 31])
def test_reference_keccak_kats_through_the_gpu_precompile(product, isa, unalignment):
    """The reference's own 8 keccak256 cases (0 / 50 / 136 / 200 bytes of 0x7b at byte misalignment 0 and 31, preceded
    by 0xff filler — src/testing/tests/precompiles/keccak256.rs:144-196) through libzkw.so: the digests the GPU
    precompile writes are asserted against the literal SURVEY Appendix C values, with no oracle in between."""
    from test_oracle_precompiles import KECCAK_KATS
    lens = (0, 50, 136, 200)
    wl = synth.make(3, isa, n_instances=70, keccak_bytes=lens, keccak_unalign=(unalignment,) * 4, sha_rounds=(1, 1, 1, 1))
    by = wl.heap_bytes
    for (start, length, _) in wl.keccak_messages:
        by[:, start - unalignment:start] = 0xFF
        by[:, start:start + length] = 123
    n, hw = by.shape[0], by.shape[1] // 32
    wl.heaps = np.ascontiguousarray(by.reshape(n, hw, 4, 8).view(">u8").reshape(n, hw, 4)[:, :, ::-1].astype("<u8"))
    bp = _run(product, wl)
    for i in (0, 1, 33, 63, 64, 69):
        t = bp.trace(i)
        assert int(t["status"]) in (K.STATUS_RUNNING, K.STATUS_ENDED)
        writes = [q for q in t["mem"] if (q["meta"] >> K.MQ_KIND_SHIFT) == 2]
        kec = writes[4:8]  # the four sha256 results come first
        assert len(kec) == 4
        for length, q in zip(lens, kec):
            assert K.u256_to_int(q["value"]).to_bytes(32, "big").hex() == KECCAK_KATS[length], (i, length, unalignment)


def test_reference_ecrecover_vectors_through_the_gpu_precompile(product, isa):
    """The two literal vectors of the reference's ecrecover test (src/testing/tests/precompiles/ecrecover.rs:128-143)
    through libzkw.so: ok marker 1 and the expected address, asserted directly."""
    from test_oracle_precompiles import ECRECOVER_VECTORS
    sigs = []
    for raw, _ in ECRECOVER_VECTORS:
        b = bytes.fromhex(raw)
        h, v, r, s = (int.from_bytes(b[i:i + 32], "big") for i in (0, 32, 64, 96))
        v = {27: 0, 28: 1, 0: 0, 1: 1}[v]  # ecrecover.rs:107-117
        sigs.append([h, r, s, v] if int(isa.consts["ecrecover_input_layout"]) == 0 else [h, v, r, s])
    wl = synth.ecrecover_workload(isa, [sigs] * 66)
    bp = _run(product, wl)
    for i in (0, 7, 63, 64, 65):
        writes = [q for q in bp.trace(i)["mem"] if (q["meta"] >> K.MQ_KIND_SHIFT) == 2]
        assert len(writes) == 4
        for j, (_, address) in enumerate(ECRECOVER_VECTORS):
            marker, word = writes[2 * j], writes[2 * j + 1]
            assert K.u256_to_int(marker["value"]) == 1
            w = K.u256_to_int(word["value"]).to_bytes(32, "big")
            assert w[:12] == bytes(12) and w[12:].hex() == address


# ---------------------------------------------------------------------------------------------------------------
# memory after the run (zkw_batch_get_page = SimpleMemory::dump_page_content_as_u256_words, memory.rs:316-396) and
# reuse of arena slots (memory.rs:660-758) — the 64-lane counterpart of the cases in tests/test_emu_parity.py
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,kw", [(2, dict(n_instances=200)), (4, dict(n_instances=130, n_cycles=1024))])
def test_pages_after_the_run(oracle, product, isa, cfg, kw):
    from test_emu_parity import compare_pages
    wl = synth.make(cfg, isa, **kw)
    wl.bootloader_calldata = synth.Xoshiro(0xCA11DA7A, wl.n_instances).words(5)
    bo, bp = _run(oracle, wl), _run(product, wl)
    assert compare_pages(bo, bp, wl, list(range(0, wl.n_instances, 13)) + [wl.n_instances - 1]) > 50
    page = int(isa.consts["bootloader_calldata_page"])
    assert np.array_equal(bp.page(7, page, 0, 5), wl.bootloader_calldata[7])
    bo.destroy()
    bp.destroy()


@pytest.mark.parametrize("lanes", [0, 64, 8])
def test_arena_slots_are_reused(oracle, product, isa, lanes):
    """64 sequential far calls under max_far_frames = 4 (two live returndata pages + the bootloader frame: one slot is
    reused some sixty times), bit-exact, and every page the run touched reads back like the reference's"""
    from test_emu_parity import compare_pages
    wl = synth.many_far_calls(isa, n_calls=64, n_instances=150)
    bo, bp = _run(oracle, wl), _run(product, wl, lanes)
    for i in range(wl.n_instances):
        tp = bp.trace(i)
        assert tp["status"] == K.STATUS_RUNNING and int(np.sum(tp["aux"]["type"] == K.AUX_FRAME_START)) == 65
        ok, why = K.traces_equal(bo.trace(i), tp)
        assert ok, "instance %d: %s" % (i, why)
    assert compare_pages(bo, bp, wl, [0, 63, 64, 149]) > 50
    bo.destroy()
    bp.destroy()


@pytest.mark.parametrize("lanes", [0, 8])
def test_decommits_are_not_capped_by_the_frame_limit(oracle, product, isa, lanes):
    """64 distinct code hashes under max_far_frames = 4 (decommitter.rs:38-96: the history is unbounded), the hashed
    known_hashes lookup and the per-pair history rows on the device; 17 decommits are repeats (not fresh)"""
    from test_emu_parity import compare_pages
    wl = synth.many_far_calls(isa, n_instances=70, distinct=64, max_far_frames=4)
    bo, bp = _run(oracle, wl), _run(product, wl, lanes)
    for i in range(wl.n_instances):
        tp = bp.trace(i)
        assert tp["status"] == K.STATUS_RUNNING
        ok, why = K.traces_equal(bo.trace(i), tp)
        assert ok, "instance %d: %s" % (i, why)
    dec = bp.trace(69)["aux"]
    dec = dec[dec["type"] == K.AUX_DECOMMIT]
    assert int(np.sum(dec["flag"] == 1)) == 66 and int(np.sum(dec["flag"] == 0)) == 17
    assert np.array_equal(bo.commitments(), bp.commitments())
    assert compare_pages(bo, bp, wl, [0, 63, 64, 69]) > 50
    bo.destroy()
    bp.destroy()


def test_arena_limit_is_a_status(product, isa):
    wl = synth.many_far_calls(isa, n_calls=6, plan="KKKKKK", n_instances=70, max_far_frames=4)
    bp = _run(product, wl)
    assert all(bp.trace(i)["status"] == K.STATUS_LIMIT for i in range(wl.n_instances))
    bp.destroy()


# ---------------------------------------------------------------------------------------------------------------
# the ISA table is an INPUT (tests/_metamorphic.py): the HIP path under tables that differ from the recalled
# default in what the absent zkevm_opcode_defs decides
# ---------------------------------------------------------------------------------------------------------------
META_WORKLOADS = {
    "cfg1": lambda isa: synth.make(1, isa, n_instances=256),
    "cfg2": lambda isa: synth.make(2, isa, n_instances=192),
    "cfg4": lambda isa: synth.make(4, isa, n_instances=96, n_cycles=1024),
    "fuzz": lambda isa: synth.fuzz_workload(isa, n_instances=128, n_ops=128, seed=0xF0AB),
    "far_calls": lambda isa: synth.many_far_calls(isa, n_calls=16, n_instances=64),
}


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg4", "fuzz"])
def test_renumbered_table_gives_the_same_witness(product, isa, name):
    """permuted variant numbering + permuted condition fields + clip_mode 1: product == oracle under that table, and (the cfg
    tapes, whose programs are the same instructions under either table) product == product under the default table once
    the opcode words are mapped back"""
    import _metamorphic as M
    from _oracle import load_oracle
    isa_v = M.renumbered(0x7AB1E + 16 * sum(map(ord, name)))
    orc_v, prod_v = load_oracle().open(isa_v), K.load_product().open(isa_v)
    try:
        wl_v = META_WORKLOADS[name](isa_v)
        b_ov, b_pv = _run(orc_v, wl_v), _run(prod_v, wl_v)
        n_ok = 0
        for i in range(wl_v.n_instances):
            tp = b_pv.trace(i)
            if name == "fuzz" and tp["status"] == K.STATUS_LIMIT:
                continue
            ok, why = K.traces_equal(b_ov.trace(i), tp)
            assert ok, "%s instance %d vs the oracle: %s" % (name, i, why)
            n_ok += 1
        assert n_ok >= wl_v.n_instances * 7 // 8
        assert np.array_equal(b_ov.commitments(), b_pv.commitments()) or name == "fuzz"
        if name != "fuzz":
            wl_d = META_WORKLOADS[name](isa)
            b_pd = _run(product, wl_d)
            for i in range(wl_d.n_instances):
                ok, why = M.same_witness(isa_v, b_pd.trace(i), b_pv.trace(i))
                assert ok, "%s instance %d vs the default table: %s" % (name, i, why)
            b_pd.destroy()
        b_ov.destroy(); b_pv.destroy()
    finally:
        orc_v.close(); prod_v.close()


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "fuzz", "far_calls"])
def test_estranged_table_matches_the_oracle(isa, name):
    """on top of the renumbering: other prices, forwarding-mode byte codes and far_call / ret register conventions"""
    import _metamorphic as M
    from _oracle import load_oracle
    isa_v = M.estranged(0xE57A + 16 * sum(map(ord, name)))
    orc_v, prod_v = load_oracle().open(isa_v), K.load_product().open(isa_v)
    try:
        wl = META_WORKLOADS[name](isa_v)
        b_o, b_p = _run(orc_v, wl), _run(prod_v, wl)
        n_ok = 0
        for i in range(wl.n_instances):
            tp = b_p.trace(i)
            if tp["status"] == K.STATUS_LIMIT:  # a capacity of the batch, a notion the reference does not have
                continue
            ok, why = K.traces_equal(b_o.trace(i), tp)
            assert ok, "%s instance %d: %s" % (name, i, why)
            n_ok += 1
        assert n_ok >= wl.n_instances * 3 // 4
        b_o.destroy(); b_p.destroy()
    finally:
        orc_v.close(); prod_v.close()


@pytest.mark.parametrize("how", ["heap", "aux", "panic"])
def test_pages_of_an_instance_that_ended(oracle, product, isa, how):
    """the bootloader's own `ret` ends the instance: stack page and the page that is not the returndata back to the pool
    (memory.rs:668-731) — zkw_batch_get_page must not see the marks of the returned frame (round-3 advisor finding)"""
    wl = synth.bootloader_returns(isa, how, n_instances=70)
    bo, bp = _run(oracle, wl), _run(product, wl)
    base = synth.BOOTLOADER_BASE_PAGE
    for i in range(wl.n_instances):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, why
        for page in (base, base + 1, base + 2, base + 3):
            assert np.array_equal(bo.page(i, page, 0, 16), bp.page(i, page, 0, 16)), (how, i, page)
    bo.destroy(); bp.destroy()


@pytest.mark.parametrize("cfg,kw,lanes", [(2, dict(n_instances=200), 0), (2, dict(n_instances=77), 8), (4, dict(n_instances=96, n_cycles=1024), 0), (1, dict(n_instances=64), 1)])
def test_expand_records_on_the_device(product, isa, cfg, kw, lanes):
    """zkw_batch_expand_records: the 512-byte CycleRecords written by the streaming kernel into a device buffer are the
    records zkw_batch_get_instance_trace rebuilds on the host (which the other tests compare with the oracle), for a sub-range
    of the instances that starts and ends inside a wave; nothing is written beyond an instance's last cycle or outside the range"""
    import torch
    wl = synth.make(cfg, isa, **kw)
    b = _run(product, wl, lanes)
    first, count = 3, wl.n_instances - 7
    stride = wl.n_cycles + 1
    dst = torch.full((count, stride, 512), 0xAB, dtype=torch.uint8, device="cuda")
    b.expand_records(first, count, dst.data_ptr(), stride, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = dst.cpu().numpy()
    for i in list(range(first, first + count, 5)) + [first + count - 1]:
        t = b.trace(i)
        n = t["n_cycles"]
        assert host[i - first, :n].tobytes() == t["records"].tobytes(), "instance %d" % i
        assert (host[i - first, n:] == 0xAB).all()
    # cycle-major layout (the records of a cycle contiguous)
    cm = torch.full((stride, count, 512), 0xEE, dtype=torch.uint8, device="cuda")
    b.expand_records(first, count, cm.data_ptr(), 1, torch.cuda.current_stream().cuda_stream, cycle_stride=count)
    torch.cuda.synchronize()
    hcm = cm.cpu().numpy()
    for i in list(range(first, first + count, 5)) + [first + count - 1]:
        t = b.trace(i)
        assert hcm[:t["n_cycles"], i - first].tobytes() == t["records"].tobytes(), "cycle-major: instance %d" % i
    # the fused entry: every instance of two batches (the second with its own lane width), unchunked or chunked as the launch decides
    b2 = _run(product, synth.make(cfg, isa, **kw), 0)
    outs = [torch.full((wl.n_instances, stride, 512), 0xCD, dtype=torch.uint8, device="cuda") for _ in range(2)]
    product.expand_records_many([b, b2], [o.data_ptr() for o in outs], stride, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for o in outs:
        host = o.cpu().numpy()
        for i in list(range(0, wl.n_instances, 7)) + [wl.n_instances - 1]:
            t = b.trace(i)
            assert host[i, :t["n_cycles"]].tobytes() == t["records"].tobytes(), "fused: instance %d" % i
    b.destroy(); b2.destroy()


RAGGED = {
    "fuzz": lambda isa: synth.fuzz_workload(isa, n_instances=200, n_ops=96, seed=0xF0E1),
    "far_calls": lambda isa: synth.many_far_calls(isa, n_calls=12, n_instances=70),
    "bootloader_returns": lambda isa: synth.bootloader_returns(isa, "heap", n_instances=70),
    "arena_limit": lambda isa: synth.many_far_calls(isa, n_calls=6, plan="KKKKKK", n_instances=70, max_far_frames=4),
}


@pytest.mark.parametrize("name,lanes", [("fuzz", 0), ("fuzz", 8), ("far_calls", 0), ("bootloader_returns", 0), ("arena_limit", 0)])
def test_expand_records_of_ragged_waves(product, isa, name, lanes):
    """The device path of zkw_expand_kernel (the software-pipelined applying wave + the streaming waves with their kmin / kmax
    branch) on waves whose lanes run DIFFERENT numbers of cycles: instances that ended early, failed lanes (the far-call plan
    that overruns its arena: ZKW_STATUS_LIMIT), fuzz tapes with several dirty-mask bits per cycle and lanes dropping out —
    a lone chunked batch and a fused call, both layouts (round-4 advisor finding: only the single-thread emulation branch had
    seen these)."""
    import torch
    wl = RAGGED[name](isa)
    b = _run(product, wl, lanes)
    ncy = [b.trace(i)["n_cycles"] for i in range(wl.n_instances)]
    if name in ("fuzz", "bootloader_returns", "arena_limit"):
        assert min(ncy) < wl.n_cycles, "the workload was meant to leave some instances short"
    stride = wl.n_cycles + 1
    sp = torch.cuda.current_stream().cuda_stream
    first, count = 1, wl.n_instances - 2
    dst = torch.full((count, stride, 512), 0xAB, dtype=torch.uint8, device="cuda")
    b.expand_records(first, count, dst.data_ptr(), stride, sp)  # a lone batch: chunked
    cm = torch.full((stride, count, 512), 0xEE, dtype=torch.uint8, device="cuda")
    b.expand_records(first, count, cm.data_ptr(), 1, sp, cycle_stride=count)
    torch.cuda.synchronize()
    host, hcm = dst.cpu().numpy(), cm.cpu().numpy()
    for i in range(first, first + count):
        t = b.trace(i)
        n = t["n_cycles"]
        assert host[i - first, :n].tobytes() == t["records"].tobytes(), "%s instance %d (%d cycles)" % (name, i, n)
        assert (host[i - first, n:] == 0xAB).all(), "%s instance %d: written behind its last cycle" % (name, i)
        assert hcm[:n, i - first].tobytes() == t["records"].tobytes(), "cycle-major: instance %d" % i
        assert (hcm[n:, i - first] == 0xEE).all()
    b2 = _run(product, RAGGED[name](isa), 0)
    outs = [torch.full((wl.n_instances, stride, 512), 0xCD, dtype=torch.uint8, device="cuda") for _ in range(2)]
    product.expand_records_many([b, b2], [o.data_ptr() for o in outs], stride, sp)  # fused
    torch.cuda.synchronize()
    for o, bb in zip(outs, (b, b2)):  # (each against its OWN batch: an instance that overruns a stream capacity stops where its wave's pooled capacity ends, which depends on the lane width)
        host = o.cpu().numpy()
        for i in range(wl.n_instances):
            t = bb.trace(i)
            assert host[i, :t["n_cycles"]].tobytes() == t["records"].tobytes(), "fused: instance %d" % i
            assert (host[i, t["n_cycles"]:] == 0xCD).all()
    b.destroy(); b2.destroy()


# ---------------------------------------------------------------------------------------------------------------
# the host-delivery path (zkw_delivery, include/zkw.h): whole steps through the 256-thread pack kernel into the pinned ring,
# traces rebuilt from the ring == oracle; fresh inputs through zkw_batch_restage
# ---------------------------------------------------------------------------------------------------------------
DELIVERY_WORKLOADS = {
    "cfg2": lambda isa: synth.make(2, isa, n_instances=200),
    "cfg4": lambda isa: synth.make(4, isa, n_instances=96, n_cycles=512),
    "cfg3": lambda isa: synth.make(3, isa, n_instances=70, keccak_bytes=(136, 300, 40, 272), keccak_unalign=(0, 31, 7, 1), sha_rounds=(1, 2, 1, 3)),
    "fuzz": lambda isa: synth.fuzz_workload(isa, n_instances=150, n_ops=96, seed=0xF0D1),
    "far_calls": lambda isa: synth.many_far_calls(isa, n_calls=12, n_instances=70),
    "ended": lambda isa: synth.bootloader_returns(isa, "heap", n_instances=70),
}


@pytest.mark.parametrize("names,lanes,threads", [(["cfg2", "cfg4", "far_calls"], None, 8), (["fuzz", "cfg3", "ended"], None, 3), (["cfg2", "fuzz"], 8, 5)])
def test_traces_rebuilt_from_the_ring_equal_the_oracle(oracle, product, isa, names, lanes, threads):
    """zkw_delivery: one pack kernel per step writes the used extents of every wave of every batch into a pinned slot; every
    trace rebuilt from the ring == the oracle's == zkw_batch_get_instance_trace; the multi-threaded replay hands over every
    cycle once (count + order-independent checksum == the fold over the traces)"""
    from test_delivery import check_delivered_step
    info = check_delivered_step(oracle, product, isa, names, threads, lanes=lanes, workloads=DELIVERY_WORKLOADS, sample=[0, 1, 7, 8, 63, 64, 65, 95, 127, 128, 149])
    assert info["pack_ms"] > 0


def test_end_to_end_pipeline_on_the_device(oracle, product, isa):
    """bench.py's `end_to_end` loop with every sampled result checked: fresh inputs restaged on side streams, the run and the pack
    kernel on the main stream, a ring with fewer slots than groups, tickets consumed two submissions later — no restage and
    no slot reuse reaches a ticket that is still being read (tests/test_delivery.py: check_end_to_end_pipeline)"""
    import torch
    from test_delivery import check_end_to_end_pipeline
    main = torch.cuda.Stream()
    sides = [torch.cuda.Stream() for _ in range(3)]
    evs = [torch.cuda.Event() for _ in range(3)]
    check_end_to_end_pipeline(oracle, product, isa, n_instances=200, n_groups=3, per_group=2, iterations=8, host_threads=6, sample=[0, 1, 63, 64, 127, 128, 199],
                              streams=(main, sides, evs))


def test_restage_gives_fresh_inputs_on_the_device(oracle, product, isa):
    """zkw_batch_restage on a side stream: new register files, scalars, callstack rows and heap images arrive by H2D copies from
    the batch's pinned staging, the restore follows on the same stream, the run on another one is ordered behind it by an event"""
    import torch
    wl_a = synth.make(2, isa, n_instances=200)
    b = product.create_batch(wl_a)
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    ev = torch.cuda.Event()
    for seed in (0x5EED7700, 0x5EED7701, 0x5EED7702):
        wl_b = synth.make(2, isa, n_instances=200, seed=seed)
        b.restage(wl_b.states, wl_b.heaps, side.cuda_stream)
        ev.record(side)
        main.wait_event(ev)
        b.run(wl_a.n_cycles, main.cuda_stream)
        b.sync()
        wl_ref = synth.make(2, isa, n_instances=200)
        wl_ref.states, wl_ref.heaps = wl_b.states, wl_b.heaps
        bo = _run(oracle, wl_ref)
        for i in (0, 1, 63, 64, 127, 199):
            ok, why = K.traces_equal(bo.trace(i), b.trace(i))
            assert ok, "seed %x instance %d: %s" % (seed, i, why)
        bo.destroy()
    b.destroy()


# ---- round 6: the host paths that changed without a GPU to run them on (bodies shared with tests/test_delivery.py, where they run on
# both CPU emulation builds) ----
def test_reads_travel_without_their_values_on_gpu(oracle, product, isa):
    """link format 2 through the 256-thread pack kernel on the device: memory reads without values, rebuilt from the shadow memory ==
    the oracle, for every workload and both settings of ZKW_OPT_READ_VALUES"""
    from test_delivery import check_reads_travel_without_their_values
    check_reads_travel_without_their_values(oracle, product, isa)


def test_restage_after_a_ragged_upload_on_gpu(oracle, product, isa):
    from test_delivery import check_restage_after_a_ragged_upload
    check_restage_after_a_ragged_upload(oracle, product, isa)


def test_staging_ring_on_gpu(oracle, product, isa):
    from test_delivery import check_staging_ring
    check_staging_ring(oracle, product, isa)


def test_a_ticket_outlives_the_restage_and_the_destruction_of_its_batch_on_gpu(oracle, product, isa):
    from test_delivery import check_a_ticket_outlives_the_restage_and_the_destruction_of_its_batch
    check_a_ticket_outlives_the_restage_and_the_destruction_of_its_batch(oracle, product, isa)
