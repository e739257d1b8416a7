"""Pins the oracle's U256 routines (oracle/u256.hpp) against Python integers — the semantics of
ethereum_types::U256 the reference ALU relies on (add.rs:35, sub.rs:35, mul.rs:35, div.rs:50,
shift.rs:48-62)."""
import ctypes as C
import random

import numpy as np

from era_zk_evm_amd import capi as K

M = (1 << 256) - 1


def _op(lib, op, a, b):
    aa, bb = K.u256_from_int(a), K.u256_from_int(b)
    out = np.zeros((2, 4), dtype="<u8")
    rc = lib.zkwo_u256_op(C.c_int(op), aa.ctypes.data_as(C.c_void_p), bb.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return K.u256_to_int(out[0]), K.u256_to_int(out[1])


def _interesting(rng):
    kind = rng.randrange(8)
    if kind == 0:
        return rng.getrandbits(256)
    if kind == 1:
        return rng.getrandbits(rng.randrange(1, 257))
    if kind == 2:
        return M - rng.getrandbits(rng.randrange(1, 65))
    if kind == 3:
        return 1 << rng.randrange(256)
    if kind == 4:
        return (1 << rng.randrange(1, 257)) - 1
    if kind == 5:
        return rng.getrandbits(64) << (64 * rng.randrange(4))
    if kind == 6:
        return rng.choice([0, 1, 2, M, M - 1, 1 << 255, 1 << 128, (1 << 128) - 1, 1 << 64, (1 << 64) - 1])
    return rng.getrandbits(192) | (0xFFFFFFFFFFFFFFFF << 192)


def test_add_sub_mul_div_shift(oracle):
    lib = oracle.lib
    rng = random.Random(1234)
    for _ in range(4000):
        a, b = _interesting(rng), _interesting(rng)
        r, of = _op(lib, 0, a, b)
        assert r == (a + b) & M and of == int(a + b > M)
        r, of = _op(lib, 1, a, b)
        assert r == (a - b) & M and of == int(a < b)
        lo, hi = _op(lib, 2, a, b)
        assert lo == (a * b) & M and hi == (a * b) >> 256
        if b:
            q, rem = _op(lib, 3, a, b)
            assert (q, rem) == divmod(a, b), (hex(a), hex(b))
        n = rng.choice([0, 1, 7, 8, 63, 64, 65, 127, 128, 255, 256, rng.randrange(257)])
        assert _op(lib, 4, a, n)[0] == (a << n) & M
        assert _op(lib, 5, a, n)[0] == a >> n


def test_knuth_d_corner_cases(oracle):
    lib = oracle.lib
    B = 1 << 64
    cases = [
        (M, 1), (M, M), (M, M - 1), (M - 1, M), ((1 << 255), (1 << 128) + 1), (M, (1 << 128) - 1),
        # qhat over-estimates / add-back step
        ((0x8000000000000000 << 192) | (0xFFFFFFFFFFFFFFFE << 128), (0x8000000000000000 << 64) | 0xFFFFFFFFFFFFFFFF),
        ((B**3 - 1) * B, B**2 - 1), (B**4 - B**2, B**2 + B - 1), ((B // 2) * B**3, (B // 2) * B + 1), (B**3, B**2 - 1),
        (0x7FFFFFFFFFFFFFFF_8000000000000000_0000000000000000_0000000000000000, 0x8000000000000000_0000000000000001),
    ]
    for a, b in cases:
        assert _op(lib, 3, a, b) == divmod(a, b), (hex(a), hex(b))
