// Keccak-f[1600] on a state whose lanes 1, 2, 8, 12, 17, 20 are held complemented (the "lane complementing transform" of the
// Keccak implementation overview, 2.2): chi then needs one NOT per plane instead of five
ZD void zk_keccak_f1600_lc(u64 a[25]) {
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
    u64 t = a[1], b;
#define ZK_RP(j, r) b = a[j]; a[j] = zk_rotl64(t, r); t = b;
    ZK_RP(10, 1) ZK_RP(7, 3) ZK_RP(11, 6) ZK_RP(17, 10) ZK_RP(18, 15) ZK_RP(3, 21) ZK_RP(5, 28) ZK_RP(16, 36) ZK_RP(8, 45) ZK_RP(21, 55) ZK_RP(24, 2)
    ZK_RP(4, 14) ZK_RP(15, 27) ZK_RP(23, 41) ZK_RP(19, 56) ZK_RP(13, 8) ZK_RP(12, 25) ZK_RP(2, 43) ZK_RP(20, 62) ZK_RP(14, 18) ZK_RP(22, 39)
    ZK_RP(9, 61) ZK_RP(6, 20) ZK_RP(1, 44)
#undef ZK_RP
    {
      const u64 B0 = a[0], B1 = a[1], B2 = a[2], B3 = a[3], B4 = a[4];
      a[0] = B0 ^ (B1 | B2); a[1] = B1 ^ (~B2 | B3); a[2] = B2 ^ (B3 & B4); a[3] = B3 ^ (B4 | B0); a[4] = B4 ^ (B0 & B1);
    }
    {
      const u64 B0 = a[5], B1 = a[6], B2 = a[7], B3 = a[8], B4 = a[9];
      a[5] = B0 ^ (B1 | B2); a[6] = B1 ^ (B2 & B3); a[7] = B2 ^ (B3 | ~B4); a[8] = B3 ^ (B4 | B0); a[9] = B4 ^ (B0 & B1);
    }
    {
      const u64 B0 = a[10], B1 = a[11], B2 = a[12], B3 = a[13], B4 = a[14], n3 = ~B3;
      a[10] = B0 ^ (B1 | B2); a[11] = B1 ^ (B2 & B3); a[12] = B2 ^ (n3 & B4); a[13] = n3 ^ (B4 | B0); a[14] = B4 ^ (B0 & B1);
    }
    {
      const u64 B0 = a[15], B1 = a[16], B2 = a[17], B3 = a[18], B4 = a[19], n3 = ~B3;
      a[15] = B0 ^ (B1 & B2); a[16] = B1 ^ (B2 | B3); a[17] = B2 ^ (n3 | B4); a[18] = n3 ^ (B4 & B0); a[19] = B4 ^ (B0 | B1);
    }
    {
      const u64 B0 = a[20], B1 = a[21], B2 = a[22], B3 = a[23], B4 = a[24], n1 = ~B1;
      a[20] = B0 ^ (n1 & B2); a[21] = n1 ^ (B2 | B3); a[22] = B2 ^ (B3 & B4); a[23] = B3 ^ (B4 | B0); a[24] = B4 ^ (B0 & B1);
    }
    a[0] ^= ZKW_KECCAK_RC[round];
  }
}
