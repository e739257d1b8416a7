"""Independent pure-Python (arbitrary precision ints) ecrecover used only to check the oracle / product
implementations of the ecrecover precompile; keccak256 comes from the already pinned oracle KATs."""
P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8


def _add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def _mul(k, pt):
    acc = None
    while k:
        if k & 1:
            acc = _add(acc, pt)
        pt = _add(pt, pt)
        k >>= 1
    return acc


def recover_pubkey(digest, r, s, v):
    """-> (x, y) or None.  digest, r, s: ints; v: 0/1 (parity of R.y)"""
    if not (0 < r < N and 0 < s < N):
        return None
    y2 = (pow(r, 3, P) + 7) % P
    y = pow(y2, (P + 1) // 4, P)
    if y * y % P != y2:
        return None
    if (y & 1) != (v & 1):
        y = P - y
    z = digest % N
    rinv = pow(r, -1, N)
    q = _add(_mul((-z * rinv) % N, (GX, GY)), _mul(s * rinv % N, (r, y)))
    return q


def keccak256(data):
    RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001, 0x8000000080008081,
          0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B,
          0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A, 0x8000000080008081,
          0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
    ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
    M = (1 << 64) - 1
    rol = lambda x, n: ((x << n) | (x >> (64 - n))) & M if n else x
    st = [[0] * 5 for _ in range(5)]
    data = bytearray(data)
    data.append(0x01)
    while len(data) % 136:
        data.append(0)
    data[-1] |= 0x80
    for off in range(0, len(data), 136):
        for i in range(17):
            st[i % 5][i // 5] ^= int.from_bytes(data[off + 8 * i: off + 8 * i + 8], "little")
        for rnd in range(24):
            c = [st[x][0] ^ st[x][1] ^ st[x][2] ^ st[x][3] ^ st[x][4] for x in range(5)]
            d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
            st = [[st[x][y] ^ d[x] for y in range(5)] for x in range(5)]
            b = [[0] * 5 for _ in range(5)]
            for x in range(5):
                for y in range(5):
                    b[y][(2 * x + 3 * y) % 5] = rol(st[x][y], ROT[x][y])
            st = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
            st[0][0] ^= RC[rnd]
    out = b"".join(st[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


def ecrecover_address(digest, r, s, v):
    q = recover_pubkey(digest, r, s, v)
    if q is None:
        return None
    return keccak256(q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big"))[12:]
