"""ecrecover precompile through the whole VM path: emulated product kernel vs oracle (bit-exact traces) and both vs
the independent Python implementation (tests/secp256k1_ref.py)."""
import os
import random
import sys

import numpy as np
import pytest

from era_zk_evm_amd import capi as K, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import secp256k1_ref as S  # noqa: E402


def make_signatures(n_instances, n_sigs, seed=0xEC0, layout=0, spoil=True):
    """valid signatures from random keys (+ a few deliberately invalid ones); returns (words, expected addresses)"""
    rng = random.Random(seed)
    words, expect = [], []
    for i in range(n_instances):
        wi, ei = [], []
        for j in range(n_sigs):
            d = rng.getrandbits(256) % (S.N - 1) + 1
            k = rng.getrandbits(256) % (S.N - 1) + 1
            h = rng.getrandbits(256)
            R = S._mul(k, (S.GX, S.GY))
            r = R[0] % S.N
            s = pow(k, -1, S.N) * (h % S.N + r * d) % S.N
            v = R[1] & 1
            kind = rng.randrange(6) if spoil else 0
            if kind == 4:
                s = 0                       # out-of-range scalar
            elif kind == 5:
                r = (r + 1) % S.N           # most likely not the x of a curve point, or another key
            addr = S.ecrecover_address(h, r, s, v)
            if kind == 0 and r == R[0]:     # untouched signature: the recovered key is the signer's
                pub = S._mul(d, (S.GX, S.GY))
                assert addr == S.keccak256(pub[0].to_bytes(32, "big") + pub[1].to_bytes(32, "big"))[12:]
            wi.append((h, r, s, v) if layout == 0 else (h, v, r, s))
            ei.append(addr)
        words.append(wi)
        expect.append(ei)
    return words, expect


@pytest.fixture(scope="module")
def emu(isa):
    import build_emu
    be = K.Backend(build_emu.build(), "zkw_").open(isa)
    yield be
    be.close()


def run(backend, wl):
    b = backend.create_batch(wl)
    b.reset()
    b.run(wl.n_cycles)
    b.sync()
    return b


def check_against_python(wl, trace, expect_i):
    writes = [q for q in trace["mem"] if (q["meta"] >> K.MQ_KIND_SHIFT) == 2]
    assert len(writes) == 2 * len(expect_i)
    for j, addr in enumerate(expect_i):
        marker = K.u256_to_int(writes[2 * j]["value"])
        word = K.u256_to_int(writes[2 * j + 1]["value"]).to_bytes(32, "big")
        assert writes[2 * j]["index"] == wl.out_base + 2 * j and writes[2 * j + 1]["index"] == wl.out_base + 2 * j + 1
        if addr is None:
            assert marker == 0 and word == bytes(32)
        else:
            assert marker == 1 and word[:12] == bytes(12) and word[12:] == addr


def test_ecrecover_through_the_vm(oracle, emu, isa):
    words, expect = make_signatures(3, 3)
    wl = synth.ecrecover_workload(isa, words)
    bo, be = run(oracle, wl), run(emu, wl)
    for i in range(wl.n_instances):
        to, te = bo.trace(i), be.trace(i)
        ok, why = K.traces_equal(to, te)
        assert ok, (i, why)
        assert to["status"] == K.STATUS_RUNNING
        check_against_python(wl, te, expect[i])
        reads = [q for q in te["mem"] if (q["meta"] >> K.MQ_KIND_SHIFT) == 1]
        assert len(reads) == 4 * len(expect[i])


def test_ecrecover_evm_word_order(oracle, isa):
    """consts.ecrecover_input_layout = 1: (hash, v, r, s)"""
    isa2 = K.Isa()
    isa2.table["consts"]["ecrecover_input_layout"] = 1
    import build_emu
    from _oracle import load_oracle

    orc2 = load_oracle().open(isa2)
    emu2 = K.Backend(build_emu.build(), "zkw_").open(isa2)
    words, expect = make_signatures(2, 2, seed=7, layout=1)
    wl = synth.ecrecover_workload(isa2, words)
    bo, be = run(orc2, wl), run(emu2, wl)
    for i in range(wl.n_instances):
        ok, why = K.traces_equal(bo.trace(i), be.trace(i))
        assert ok, (i, why)
        check_against_python(wl, be.trace(i), expect[i])
    orc2.close()
    emu2.close()


def test_ecrecover_bad_recovery_id_is_a_reference_panic(oracle, emu, isa):
    words, _ = make_signatures(2, 1, seed=9, spoil=False)
    h, r, s, v = words[1][0]
    words[1][0] = (h, r, s, 2)
    wl = synth.ecrecover_workload(isa, words)
    bo, be = run(oracle, wl), run(emu, wl)
    for b in (bo, be):
        assert b.trace(0)["status"] == K.STATUS_RUNNING
        assert b.trace(1)["status"] == K.STATUS_REFERENCE_PANIC
    ok, why = K.traces_equal(bo.trace(0), be.trace(0))
    assert ok, why
    # the panicking instance: everything up to the failed cycle is identical; the state a panicking reference process
    # leaves behind is not defined
    to, te = bo.trace(1), be.trace(1)
    assert to["n_cycles"] == te["n_cycles"] == 2
    for k in K.TRACE_ARRAYS:
        assert to[k].tobytes() == te[k].tobytes(), k
