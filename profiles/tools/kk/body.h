// Keccak-f[1600], state in 25 statically indexed u64
ZD void zk_keccak_f1600(u64 a[25]) {
  for (int round = 0; round < 24; round++) {
    u64 c[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; x++) {
      const u64 dd = c[(x + 4) % 5] ^ zk_rotl64(c[(x + 1) % 5], 1);
#pragma unroll
      for (int y = 0; y < 25; y += 5) a[y + x] ^= dd;
    }
    // rho + pi (explicit chain, static indices)
    u64 t = a[1], b;
#define ZK_RP(j, r) b = a[j]; a[j] = zk_rotl64(t, r); t = b;
    ZK_RP(10, 1) ZK_RP(7, 3) ZK_RP(11, 6) ZK_RP(17, 10) ZK_RP(18, 15) ZK_RP(3, 21) ZK_RP(5, 28) ZK_RP(16, 36) ZK_RP(8, 45) ZK_RP(21, 55) ZK_RP(24, 2)
    ZK_RP(4, 14) ZK_RP(15, 27) ZK_RP(23, 41) ZK_RP(19, 56) ZK_RP(13, 8) ZK_RP(12, 25) ZK_RP(2, 43) ZK_RP(20, 62) ZK_RP(14, 18) ZK_RP(22, 39)
    ZK_RP(9, 61) ZK_RP(6, 20) ZK_RP(1, 44)
#undef ZK_RP
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
      u64 r0 = a[y], r1 = a[y + 1], r2 = a[y + 2], r3 = a[y + 3], r4 = a[y + 4];
      a[y] = r0 ^ (~r1 & r2);
      a[y + 1] = r1 ^ (~r2 & r3);
      a[y + 2] = r2 ^ (~r3 & r4);
      a[y + 3] = r3 ^ (~r4 & r0);
      a[y + 4] = r4 ^ (~r0 & r1);
    }
    a[0] ^= ZKW_KECCAK_RC[round];
  }
}
