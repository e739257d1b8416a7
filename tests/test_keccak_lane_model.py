"""The lane-parallel Keccak-f[1600] of the helper waves (zkw_kh_serve), restated lane by lane in Python: one
lane per 32-bit half of a state word, every cross-lane term a fetch from another lane, rotations as 32-bit funnel
shifts whose operands are chosen by the fetch addresses.  This pins the fetch tables, the rho offsets and the funnel
operand rule the device code uses (same formulas) against hashlib's SHA3, whose permutation is the same; the device
code itself is pinned on the GPU (tests/test_gpu_parity.py: *_helper_waves)."""
import hashlib
import random

M32 = (1 << 32) - 1
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
      0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
      0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
      0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
RHO = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]  # ZKW_KH_RHO, index x + 5y


def alignbit(a, b, s):  # v_alignbit_b32: the low 32 bits of (a:b) >> (s mod 32)
    return (((a << 32) | b) >> (s & 31)) & M32


def lane_parallel_f1600(words):
    """words[i], i = x + 5y, 64-bit -> the permuted state; state[h][i] is what lane 32 h + i holds"""
    st = [[w & M32 for w in words], [w >> 32 for w in words]]
    X = [i % 5 for i in range(25)]
    Y = [i // 5 for i in range(25)]
    xm1 = [Y[i] * 5 + (X[i] + 4) % 5 for i in range(25)]
    xp1 = [Y[i] * 5 + (X[i] + 1) % 5 for i in range(25)]
    # the three theta outputs destination d needs: pi source of columns x, x + 1, x + 2 of its plane
    src = [[((X[d] + k) % 5 + 3 * Y[d]) % 5 + 5 * ((X[d] + k) % 5) for d in range(25)] for k in range(3)]
    for rnd in range(24):
        c = [[st[h][i] ^ st[h][(i + 5) % 25] ^ st[h][(i + 10) % 25] ^ st[h][(i + 15) % 25] ^ st[h][(i + 20) % 25] for i in range(25)] for h in range(2)]
        e = [[st[h][i] ^ c[h][xm1[i]] ^ alignbit(c[h][xp1[i]], c[1 - h][xp1[i]], 31) for i in range(25)] for h in range(2)]
        nxt = [[0] * 25, [0] * 25]
        for h in range(2):
            for d in range(25):
                t = []
                for k in range(3):
                    s = src[k][d]
                    amount = RHO[s]
                    swap = amount >= 32 or amount == 0  # (0 is taken as 64: halves exchanged, funnel shift 0)
                    first, second = (e[1 - h][s], e[h][s]) if swap else (e[h][s], e[1 - h][s])
                    t.append(alignbit(first, second, (32 - (amount & 31)) & 31))
                v = t[0] ^ (~t[1] & t[2] & M32)
                if d == 0:
                    v ^= (RC[rnd] >> 32) if h else (RC[rnd] & M32)
                nxt[h][d] = v & M32
        st = nxt
    return [st[0][i] | (st[1][i] << 32) for i in range(25)]


def sha3_256_with(perm, msg):
    rate = 136
    state = [0] * 25
    p = bytearray(msg) + b"\x06"
    p += b"\x00" * ((-len(p)) % rate)
    p[-1] |= 0x80
    for off in range(0, len(p), rate):
        for i in range(17):
            state[i] ^= int.from_bytes(p[off + 8 * i:off + 8 * i + 8], "little")
        state = perm(state)
    return b"".join(state[i].to_bytes(8, "little") for i in range(4))


def test_lane_parallel_permutation_is_keccak_f1600():
    rng = random.Random(0x5EED)
    for n in (0, 1, 136, 137, 300):
        msg = bytes(rng.randrange(256) for _ in range(n))
        assert sha3_256_with(lane_parallel_f1600, msg) == hashlib.sha3_256(msg).digest(), n


def test_rho_offsets_are_the_triangular_numbers():
    # rho[x][y] = (t + 1)(t + 2) / 2 mod 64 along (x, y) -> (y, 2x + 3y), starting at (1, 0) with t = 0
    x, y = 1, 0
    for t in range(24):
        assert RHO[x + 5 * y] == ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    assert RHO[0] == 0


def test_device_table_is_this_table():
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "era-zk_evm_amd", "csrc", "zkw_kernels.hip")).read()
    m = re.search(r"ZKW_KH_RHO\[32\] = \{([^}]*)\}", src)
    assert m, "ZKW_KH_RHO not found in zkw_kernels.hip"
    dev = [int(t) for t in m.group(1).replace("\n", " ").split(",")]
    assert dev[:25] == RHO and dev[25:] == [0] * 7
