"""The field helpers of the permutation (era-zk_evm_amd/csrc/zkw_goldilocks.hip.h) against big-integer arithmetic.

Inside the permutation values are only kept congruent (any u64, possibly >= p), and the linear layers run on the 32-bit
halves of the state with bounds argued in the header's comments; random inputs practically never reach those bounds, so
this test feeds the extremes (all-ones words, p - 1, p, 2^64 - 1, single-bit patterns) next to random states.  The
header is compiled by g++ in its portable form (the same expressions the device build evaluates with v_mad_u64_u32);
the device code itself is covered by the digests of the `-m gpu` parity tests."""
import ctypes as C
import os
import random
import subprocess

import pytest

P = 0xFFFFFFFF00000001
M64 = (1 << 64) - 1
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]
EXTREMES = [0, 1, 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, P - 1, P, P + 1, M64 - 1, M64, 0x8000000000000000,
            0x7FFFFFFFFFFFFFFF, 0x00000000FFFFFFFE, 0xFFFFFFFE00000000, 0xFFFFFFFEFFFFFFFF]


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(HERE, "emu", "gl_layers_probe.cpp")
    out = os.path.join(HERE, "emu", "libgl_layers_probe.so")
    hdr = os.path.join(ROOT, "era-zk_evm_amd", "csrc", "zkw_goldilocks.hip.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", os.path.join(HERE, "emu"),
                        "-I", os.path.join(ROOT, "era-zk_evm_amd", "csrc"), "-o", out, src], check=True)
    l = C.CDLL(out)
    for f in ("t_fold", "t_mulred", "t_add_rc"):
        getattr(l, f).restype = C.c_uint64
        getattr(l, f).argtypes = [C.c_uint64, C.c_uint64]
    l.t_pow7.restype = C.c_uint64
    l.t_pow7.argtypes = [C.c_uint64]
    return l


def states(rng, n_random=200):
    out = [[e] * 12 for e in EXTREMES]
    for _ in range(200):
        out.append([rng.choice(EXTREMES) for _ in range(12)])
    for _ in range(n_random):
        out.append([rng.getrandbits(64) for _ in range(12)])
    return out


def external_ref(s):
    o = []
    for b in range(0, 12, 4):
        o += [sum(M4[r][c] * s[b + c] for c in range(4)) for r in range(4)]
    sums = [o[j] + o[4 + j] + o[8 + j] for j in range(4)]
    return [(o[i] + sums[i & 3]) % P for i in range(12)]


def internal_ref(s):
    t = sum(s)
    return [(t + (s[i] << i)) % P for i in range(12)]


def call_layer(fn, s):
    a = (C.c_uint64 * 12)(*s)
    fn(a)
    return list(a)


def test_linear_layers_at_the_extremes(lib):
    rng = random.Random(7)
    for s in states(rng):
        got = call_layer(lib.t_external, s)
        assert all(g <= M64 for g in got) and [g % P for g in got] == external_ref(s), ("external", s)
        got = call_layer(lib.t_internal, s)
        assert [g % P for g in got] == internal_ref(s), ("internal", s)


def test_fold_of_halves_at_its_bounds(lib):
    """gl_fold_halves(L, H) = L + 2^32 H (mod p) on the whole range the layers can produce (L, H < 2^45) and at the
    wrap of the high word."""
    rng = random.Random(11)
    edge = [0, 1, 0xFFFFFFFF, 0x100000000, (1 << 45) - 1, (1 << 44), 0xFFFFFFFF + (0xFFF << 32), 0xFFFFFFFE, (1 << 45) - (1 << 32)]
    cases = [(l, h) for l in edge for h in edge] + [(rng.getrandbits(45), rng.getrandbits(45)) for _ in range(2000)]
    for l, h in cases:
        assert lib.t_fold(l, h) % P == (l + (h << 32)) % P, (hex(l), hex(h))


def test_sbox_and_round_constant_addition_accept_any_u64(lib):
    rng = random.Random(13)
    vals = EXTREMES + [rng.getrandbits(64) for _ in range(300)]
    for a in vals:
        assert lib.t_pow7(a) % P == pow(a, 7, P), hex(a)
        for b in EXTREMES[:8] + [rng.getrandbits(64) for _ in range(4)]:
            assert lib.t_mulred(a, b) % P == a * b % P, (hex(a), hex(b))
            rc = b % P  # constants are canonical
            assert lib.t_add_rc(a, rc) % P == (a + rc) % P, (hex(a), hex(rc))
