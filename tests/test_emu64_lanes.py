"""The lane-parallel paths of the kernels on the CPU: the `-m gpu` parity tests of tests/test_gpu_parity.py (same bodies, same
workloads, same comparisons against the oracle) run against tests/emu/libzkw_emu64.so — the product sources compiled by g++ for
64-lane waves on the SIMT engine of tests/emu/emu_simt.cpp (every lane a fiber, ballots / readlanes / stream allocations /
barriers emulated, the execution mask taken from the ZKW_DIV_* annotations of the source).  What a one-lane emulation cannot
reach is reached here without a GPU: ranks among 64 lanes, opcode-word and variant grouping of diverged lanes, the short
cycle's wave-uniform tests with lanes that fail them, thin waves with their keccak256 / decommit helper waves, the 320-thread
software pipeline of the expand kernel.  Path counters (a test hook of the emulation builds) prove that a tape really took the
path a test is named after.  Timing, LDS banking and coalescing stay with the GPU suite."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from era_zk_evm_amd import capi as K, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import test_gpu_parity as G  # noqa: E402  (the bodies; its module-level `gpu` mark stays with that module)


@pytest.fixture(scope="module")
def product(isa):
    """what the GPU tests call `product`: here the 64-lane emulation build of the same sources"""
    import build_emu
    be = K.Backend(build_emu.build(wave=64), "zkw_").open(isa)
    yield be
    be.close()


def path_counts(be, reset=True):
    """lane-cycles by path since the last reset: short cycle, of them heap accesses, general path, of them in variant groups,
    keccak256 calls served by helper waves, decommits chained by helper waves (ZKW_EMU_COUNT in zkw_kernels.hip)"""
    out = (C.c_ulonglong * 8)()
    be.lib.zkw_emu_get_path_counts(out, C.c_int(1 if reset else 0))
    return dict(zip(("short", "short_uma", "general", "variant", "kh_served", "dq_served"), list(out)[:6]))


# ---- the GPU suite's tests whose subject is lane-parallel behaviour, unchanged ----
for _name in """
test_cfg0_nop_add test_cfg1_arith_256x256 test_cfg2_mixed test_cfg2_ragged_last_wave test_cfg3_precompiles test_cfg4_l2_block
test_divergent_tapes_in_one_wave test_status_codes test_generic_per_lane_path_forced test_variant_group_path_forced
test_uniform_fuzz_shared_tape test_keccak_served_by_helper_waves test_thin_waves_without_room_for_helpers_keep_the_lane_path
test_reference_keccak_kats_through_the_helper_waves test_arena_slots_are_reused test_capacity_overruns_are_limit_statuses
test_keccak_precompile_odd_lengths_and_alignments test_queue_commitments test_rerun_after_reset_is_identical
test_split_run_equals_single_run test_net_states_nested_frames test_net_states_l2_block test_net_states_partial_run
test_register_delta_capacity_is_a_limit_status test_pages_after_the_run test_pages_of_an_instance_that_ended
test_traces_rebuilt_from_the_ring_equal_the_oracle test_reference_keccak_kats_through_the_gpu_precompile
test_reference_ecrecover_vectors_through_the_gpu_precompile test_host_replay_on_gpu test_arena_limit_is_a_status
test_decommits_are_not_capped_by_the_frame_limit
""".split():
    globals()[_name] = getattr(G, _name)


# ---- and that the paths are the ones the tests are named after ----
def _run(be, wl, lanes=0, cycles=None):
    wl.limits["lanes_per_wave"] = lanes
    b = be.create_batch(wl)
    b.reset()
    b.run(cycles or wl.n_cycles)
    b.sync()
    return b


def _equal(bo, bp, wl, what):
    for i in range(wl.n_instances):
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, "%s instance %d: %s" % (what, i, why)


def test_the_short_cycle_runs_and_refuses(oracle, product, isa):
    """cfg 2 (the headline tape) and a uniform fuzz tape on full waves: most lane-cycles go through the short cycle — heap
    accesses among them — and the rest through the general path (stack operands, far calls, lanes that parted at a
    conditional jump, an exception in one lane); with the test hook that switches the short cycle off, none does."""
    for wl in (synth.make(2, isa, n_instances=128), synth.uniform_fuzz(isa, n_instances=128, n_ops=192, seed=0xF1A0)):
        bo = _run(oracle, wl)
        path_counts(product)
        bp = _run(product, wl, 64)
        c = path_counts(product)
        _equal(bo, bp, wl, wl.name)
        total = int(bp.stats()["cycles"])
        assert c["short"] + c["general"] >= total  # (a lane masked into nop / panic visits the group loop twice)
        assert c["short"] > total // 2, c
        assert c["short_uma"] > total // 10, c
        assert c["general"] > total // 20, c
        product.set_option(K.OPT_DEBUG_FLAGS, 4)
        try:
            bq = _run(product, wl, 64)
        finally:
            product.set_option(K.OPT_DEBUG_FLAGS, 0)
        c = path_counts(product)
        _equal(bo, bq, wl, wl.name + " (per-lane groups)")
        assert c["short"] == 0 and c["general"] >= total, c
        bo.destroy(); bp.destroy(); bq.destroy()


def test_one_lane_waves_take_the_short_cycle_too(oracle, isa):
    """the one-lane emulation build (every other CPU test of the kernel logic) compiles and runs the short cycle as well"""
    import build_emu
    emu1 = K.Backend(build_emu.build(wave=1), "zkw_").open(isa)
    try:
        wl = synth.uniform_fuzz(isa, n_instances=6, n_ops=192, seed=0xF1A1)
        bo = _run(oracle, wl)
        path_counts(emu1)
        be = _run(emu1, wl)
        c = path_counts(emu1)
        _equal(bo, be, wl, wl.name)
        assert c["short"] > int(be.stats()["cycles"]) // 2 and c["short_uma"] > 0, c
    finally:
        emu1.close()


def test_short_cycle_ab_partners(oracle, isa):
    """The two A/B partners of round 6 (profiles/r10_short_cycle_census.txt; the default build is untouched by their #ifdefs).
    -DZKW_SHORT_CLASS: the short cycle reads the class of an instruction from the bits the host packs into its ISA entry instead of
    decoding it every cycle — the same cycles qualify (path counters equal to the default build's).  + -DZKW_SHORT_STACK: ALU
    instructions with stack operands (mem_ops.rs:51-121: push / pop / sp-relative / absolute, src0 read and dst0 written with their
    queries, sp moved) run in the short cycle too — more cycles qualify (the headline tape: 207 of 256 instead of 189).  Either way the
    witness is the oracle's, under the default table and under tables with other variant numbering / prices / conventions (the bits
    are packed from whatever table the caller uploads)."""
    import build_emu
    import _metamorphic as M
    from _oracle import load_oracle
    for mk in (lambda: isa, lambda: M.renumbered(0x7AB1E), lambda: M.estranged(0xE57A)):
        isa_v = mk()
        orc = oracle if isa_v is isa else load_oracle().open(isa_v)
        emu_d = K.Backend(build_emu.build(wave=1), "zkw_").open(isa_v)
        emu_c = K.Backend(build_emu.build(wave=1, defines=("ZKW_SHORT_CLASS",), tag="short_class"), "zkw_").open(isa_v)
        emu_s = K.Backend(build_emu.build(wave=1, defines=("ZKW_SHORT_CLASS", "ZKW_SHORT_STACK"), tag="short_stack"), "zkw_").open(isa_v)
        try:
            for wl in (synth.make(2, isa_v, n_instances=4), synth.uniform_fuzz(isa_v, n_instances=6, n_ops=192, seed=0xF1A2), synth.make(4, isa_v, n_instances=3, n_cycles=512)):
                bo = _run(orc, wl)
                path_counts(emu_d); bd = _run(emu_d, wl); cd = path_counts(emu_d)
                path_counts(emu_c); bc = _run(emu_c, wl); cc = path_counts(emu_c)
                path_counts(emu_s); bs = _run(emu_s, wl); cs = path_counts(emu_s)
                _equal(bo, bc, wl, wl.name + " (short class bits)")
                _equal(bo, bs, wl, wl.name + " (stack operands in the short cycle)")
                assert cc == cd and cc["short"] > 0, (cc, cd)
                assert cs["short"] > cd["short"] and cs["short"] + cs["general"] == cd["short"] + cd["general"], (cs, cd)
                if wl.name.startswith("cfg2") and isa_v is isa:
                    assert cs["short"] * 256 >= 205 * int(bs.stats()["cycles"]), cs
                bo.destroy(); bd.destroy(); bc.destroy(); bs.destroy()
        finally:
            emu_d.close(); emu_c.close(); emu_s.close()
            if isa_v is not isa:
                orc.close()


def test_diverged_lanes_form_variant_groups(oracle, product, isa):
    """Every lane of a wave runs its own program.  (a) fuzz tapes: whatever the lanes hold; (b) one sequence of opcodes with
    per-lane register numbers and immediates — the shape variant grouping exists for: the group loop widens the word groups to
    variant groups on its own (no test hook) and most lane-cycles run in zkw_vec_exec."""
    wl = synth.fuzz_workload(isa, n_instances=64, n_ops=96, seed=0xF0B1)
    bo = _run(oracle, wl)
    path_counts(product)
    bp = _run(product, wl, 64)
    c = path_counts(product)
    compared = 0
    for i in range(wl.n_instances):
        tp = bp.trace(i)
        if int(tp["status"]) == K.STATUS_LIMIT:
            continue
        ok, why = K.traces_equal(bo.trace(i), tp)
        assert ok, "instance %d: %s" % (i, why)
        compared += 1
    assert compared * 8 > wl.n_instances * 7
    assert c["short"] == 0 and c["variant"] > 0, c
    bo.destroy(); bp.destroy()
    # (b)
    n, n_ops = 64, 64
    wl = synth.make(1, isa, n_instances=n, n_cycles=n_ops)
    rng = synth.ScalarRng(0xB0B)
    kinds = [rng.below(5) for _ in range(n_ops)]
    wl.blobs, wl.code_pages = [], []
    for i in range(n):
        ops = []
        for kd in kinds:
            kw = dict(src0=rng.below(16), src1=rng.below(16), dst0=1 + rng.below(15), flags=1)
            if kd == 0: ops.append(isa.enc(K.OP_ADD, **kw))
            elif kd == 1: ops.append(isa.enc(K.OP_SUB, **kw))
            elif kd == 2: ops.append(isa.enc(K.OP_MUL, dst1=1 + rng.below(15), **kw))
            elif kd == 3: ops.append(isa.enc(K.OP_BINOP, variant=K.BINOP_XOR, **kw))
            else: ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=rng.below(1 << 16), src1=rng.below(16), dst0=1 + rng.below(15)))
        wl.blobs.append(K.pack_code(ops + [isa.enc(K.OP_NOP)] * 4))
        wl.code_pages.append((i, 1, synth.BOOTLOADER_CODE_PAGE, i))
    bo = _run(oracle, wl)
    path_counts(product)
    bp = _run(product, wl, 64)
    c = path_counts(product)
    _equal(bo, bp, wl, "per-lane registers")
    assert c["variant"] * 2 > c["general"], c
    bo.destroy(); bp.destroy()


def test_helper_waves_serve_keccak_and_decommits(oracle, product, isa):
    """thin waves: the keccak256 calls of cfg 3 go through the mailbox to the helper waves (lane-parallel sponge over
    ds_bpermute); a whole step of cfg 2 on full waves hands its decommits to the workgroup's helper wave — the digests and
    the decommit-queue commitment equal the oracle's, and the counters say who computed them"""
    wl = synth.make(3, isa, n_instances=6, keccak_k=(1, 2, 3, 1), sha_rounds=(1, 2, 3, 5))
    bo = _run(oracle, wl)
    path_counts(product)
    bp = _run(product, wl, 2)
    c = path_counts(product)
    _equal(bo, bp, wl, "cfg3 on 2-lane waves")
    assert c["kh_served"] == 4 * wl.n_instances, c
    bo.destroy(); bp.destroy()
    wl = synth.make(2, isa, n_instances=96)
    wl.limits["lanes_per_wave"] = 64
    bo = _run(oracle, wl)
    bp = product.create_batch(wl)
    path_counts(product)
    product.step_many([bp], wl.n_cycles, 1 << K.QUEUE_DECOMMIT)
    bp.sync()
    c = path_counts(product)
    _equal(bo, bp, wl, "cfg2 step")
    assert np.array_equal(bo.commitments()[:, K.QUEUE_DECOMMIT], bp.commitments()[:, K.QUEUE_DECOMMIT])
    assert c["dq_served"] == 2 * wl.n_instances, c  # (two far calls with a decommit per instance of the cfg-2 tape)
    bo.destroy(); bp.destroy()
