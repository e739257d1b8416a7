"""The 256-bit shift / window / mask helpers of the product (era-zk_evm_amd/csrc/zkw_u256.hip.h) against Python integers.

The VM-level parity tests reach these through the shift opcodes and the unaligned memory accesses; here every shift
amount 0..300 (the `uint` crate's "n >= 256 => 0", shift.rs:51,58), every byte offset 0..31 of an unaligned read
(uma.rs:291-300) and every mask width are checked directly, on patterns that expose a misplaced limb or a funnel shift
by zero.  The header is compiled by g++ against the single-lane HIP stand-in (the same source the device build compiles);
the device code is covered by the `-m gpu` parity tests."""
import ctypes as C
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
M = (1 << 256) - 1
A8 = C.c_uint32 * 8


def limbs(x):
    return A8(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def value(a):
    return sum(int(a[i]) << (32 * i) for i in range(8))


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(HERE, "emu", "u256_probe.cpp")
    out = os.path.join(HERE, "emu", "libu256_probe.so")
    hdr = os.path.join(ROOT, "era-zk_evm_amd", "csrc", "zkw_u256.hip.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", os.path.join(HERE, "emu"),
                        "-I", os.path.join(ROOT, "era-zk_evm_amd", "csrc"), "-o", out, src], check=True)
    return C.CDLL(out)


def patterns():
    rng = random.Random(21)
    distinct = sum((0x11111111 * (i + 1)) << (32 * i) for i in range(8))  # every limb and every nibble position recognisable
    return [0, 1, M, 1 << 255, distinct, 0x0123456789ABCDEF_FEDCBA9876543210_0F1E2D3C4B5A6978_8796A5B4C3D2E1F0] + [rng.getrandbits(256) for _ in range(6)]


def test_shifts_for_every_amount(lib):
    out = A8()
    for x in patterns():
        a = limbs(x)
        for n in list(range(0, 301)) + [511, 512, 1 << 16, 0xFFFFFFFF]:
            lib.t_shl(a, C.c_uint32(n), out)
            assert value(out) == ((x << n) & M if n < 256 else 0), ("shl", hex(x), n)
            lib.t_shr(a, C.c_uint32(n), out)
            assert value(out) == (x >> n if n < 256 else 0), ("shr", hex(x), n)


def test_low_mask_for_every_width(lib):
    out = A8()
    for nbits in range(0, 257):
        lib.t_low_mask(C.c_uint32(nbits), out)
        assert value(out) == (1 << nbits) - 1, nbits


def test_byte_window_for_every_offset(lib):
    """u256_byte_window(hi, lo, u) = bytes [u, u + 32) of the 64-byte big-endian string hi || lo"""
    out = A8()
    ps = patterns()
    for hi in ps[2:8]:
        for lo in ps[3:9]:
            both = (hi << 256) | lo
            for u in range(32):
                lib.t_window(limbs(hi), limbs(lo), C.c_uint32(u), out)
                assert value(out) == (both >> (256 - 8 * u)) & M, (hex(hi), hex(lo), u)


def test_uniform_offset_window_and_merge(lib):
    """u256_byte_window_at / u256_merge_at (the wave-uniform forms op_uma takes when every lane has the same offset) for every
    offset 1..31: the window equals bytes [u, u + 32) of hi || lo, the merge equals hi || lo with those 32 bytes replaced
    (uma.rs:291-300, 349-400)"""
    out, n0, n1 = A8(), A8(), A8()
    ps = patterns()
    for hi in ps[2:8]:
        for lo in ps[3:9]:
            both = (hi << 256) | lo
            for u in range(1, 32):
                lib.t_window_at(limbs(hi), limbs(lo), C.c_uint32(u), out)
                assert value(out) == (both >> (256 - 8 * u)) & M, ("window", hex(hi), hex(lo), u)
                for v in (ps[4], ps[5], ps[9]):
                    lib.t_merge_at(limbs(hi), limbs(lo), limbs(v), C.c_uint32(u), n0, n1)
                    sh = 256 - 8 * u
                    want = (both & ~(M << sh)) | (v << sh)
                    assert (value(n0) << 256) | value(n1) == want, ("merge", hex(hi), hex(lo), hex(v), u)


def test_select_bits(lib):
    out = A8()
    rng = random.Random(5)
    for _ in range(50):
        m, a, b = rng.getrandbits(256), rng.getrandbits(256), rng.getrandbits(256)
        lib.t_select_bits(limbs(m), limbs(a), limbs(b), out)
        assert value(out) == (m & a) | (~m & b & M)


def test_add_sub_mul_divmod(lib):
    """the ALU's arithmetic (add.rs:35, sub.rs:35, mul.rs:35-39, div.rs:50) on the operand classes and the Knuth-D
    corner cases the oracle's own routines are pinned with (tests/test_oracle_u256.py)"""
    from test_oracle_u256 import _interesting
    lib.t_add.restype = C.c_uint32
    lib.t_sub.restype = C.c_uint32
    rng = random.Random(4321)
    o1, o2 = A8(), A8()
    B = 1 << 64
    div_cases = [
        (M, 1), (M, M), (M, M - 1), (M - 1, M), ((1 << 255), (1 << 128) + 1), (M, (1 << 128) - 1),
        ((0x8000000000000000 << 192) | (0xFFFFFFFFFFFFFFFE << 128), (0x8000000000000000 << 64) | 0xFFFFFFFFFFFFFFFF),
        ((B**3 - 1) * B, B**2 - 1), (B**4 - B**2, B**2 + B - 1), ((B // 2) * B**3, (B // 2) * B + 1), (B**3, B**2 - 1),
        (0x7FFFFFFFFFFFFFFF_8000000000000000_0000000000000000_0000000000000000, 0x8000000000000000_0000000000000001),
    ]
    pairs = [(_interesting(rng), _interesting(rng)) for _ in range(3000)] + div_cases
    for a, b in pairs:
        of = lib.t_add(limbs(a), limbs(b), o1)
        assert value(o1) == (a + b) & M and of == int(a + b > M)
        of = lib.t_sub(limbs(a), limbs(b), o1)
        assert value(o1) == (a - b) & M and of == int(a < b)
        lib.t_mul(limbs(a), limbs(b), o1, o2)
        assert value(o1) == (a * b) & M and value(o2) == (a * b) >> 256
        if b:
            lib.t_divmod(limbs(a), limbs(b), o1, o2)
            assert (value(o1), value(o2)) == divmod(a, b), (hex(a), hex(b))
