// Precompile round functions on device: keccak256 and sha256 over the calling lane's memory.
//
// Reference call site: VmState::call_precompile (src/vm_state/helpers.rs:196-223) ->
// DefaultPrecompilesProcessor::execute_precompile (zk_evm_abstractions @ v1.4.1, not on disk; the
// restated semantics and their pinning are documented in oracle/vm.cpp and DESIGN.md §precompiles).
// Observable behaviour reproduced here: which memory words are read (each input word once, in
// order, MemoryType::FatPointer for keccak / Heap for sha256, at query.timestamp), the digest, and
// the single result write (MemoryType::Heap, at query.timestamp + 1).  One message per lane; the
// Keccak state (25 x u64) and the SHA-256 schedule stay in VGPRs with static indexing, the
// 136-byte Keccak rate block is assembled byte-wise in a per-lane LDS row.
#pragma once

__device__ const u64 ZKW_KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

__device__ const u32 ZKW_SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// rotation of a 64-bit lane by a constant: two funnel shifts of its halves (v_alignbit_b32) — left to itself the compiler
// builds it from 64-bit shifts and an or (four to five instructions, the 64-bit shifts at a fraction of the rate)
ZD u64 zk_rotl64(u64 x, int n) {
#ifdef __HIP_DEVICE_COMPILE__
  u32 lo = (u32)x, hi = (u32)(x >> 32);
  if (n >= 32) {
    const u32 t = lo;
    lo = hi;
    hi = t;
    n -= 32;
  }
  if (n == 0) return ((u64)hi << 32) | lo;
  // {a, b} >> s, low word: alignbit(hi, lo, 32 - n) = hi << n | lo >> (32 - n)
  return ((u64)__builtin_amdgcn_alignbit(hi, lo, (u32)(32 - n)) << 32) | (u64)__builtin_amdgcn_alignbit(lo, hi, (u32)(32 - n));
#else
  return (x << n) | (x >> (64 - n));
#endif
}

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

// Keccak-f[1600] on a state whose lanes 1, 2, 8, 12, 17, 20 are held complemented (the "lane complementing transform" of the
// Keccak implementation overview, 2.2): chi needs one NOT per plane instead of five, and and / or + xor (two-operand
// encodings) take the place of the three-operand v_bfi the plain form compiles to — those issue at half the rate when two
// waves share a SIMD (profiles/tools/kk: 5.9 -> 6.3 G permutations/s with the chip full).  Absorbing is an XOR and does not
// care; the state starts with those lanes all-ones and the digest un-complements lanes 1 and 2 (precompile_keccak256).
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
ZD void zk_keccak_lc_flip(u64 st[25]) { st[1] = ~st[1]; st[2] = ~st[2]; st[8] = ~st[8]; st[12] = ~st[12]; st[17] = ~st[17]; st[20] = ~st[20]; }

#define ZKW_KECCAK_RATE 136

ZD void keccak_absorb_block(Shared& sh, u32 lane, u64 st[25]) {
#pragma unroll
  for (int i = 0; i < 17; i++) {
    const u64 w = (u64)sh.krow[(2 * i) * sh.L + lane] | ((u64)sh.krow[(2 * i + 1) * sh.L + lane] << 32);
    st[i] ^= w;
  }
  zk_keccak_f1600_lc(st);
}

// big-endian dword k (0 = most significant) of a memory word, k dynamic: select chain instead of register indexing
ZD u32 word_be_dword(const u256& w, u32 k) {
  u32 v = w.w[7];
#pragma unroll
  for (int i = 1; i < 8; i++) v = k == (u32)i ? w.w[7 - i] : v;
  return v;
}

// keccak256_rounds_function: input = `input_memory_length` bytes at byte offset `input_memory_offset`
// of page `memory_page_to_read`; output = one big-endian word at word `output_memory_offset` of
// `memory_page_to_write` (reference test src/testing/tests/precompiles/keccak256.rs:99-139).
// The message is consumed four bytes at a time: stream dword d is a funnel of two consecutive big-endian memory dwords
// (the byte misalignment of the input is constant over the message); the rate block is staged as 34 dwords per lane in
// the wave's row buffer ([dword][lane], one coalesced store per dword) and absorbed with static indices.  Every memory
// word that holds message bytes is read exactly once, in order, as in the reference.
// The share of memory word J (of the up to six that a full 136-byte block reaches into: 35 dwords from dword 7 of a word) in the block whose first stream
// dword sits at dword PHI of word 0: stream dword p = 8 J + k of the word run is block dword p - PHI, and — when the
// message is not dword-aligned (`sh8` = misalignment in bits, wave-uniform) — its low bits also are the tail of block
// dword p - 1 - PHI.  The two parts of a funnelled dword occupy disjoint bytes, so each memory dword is XORed into the
// state on its own (byte-swapped into the little-endian lanes): no word has to wait for its neighbour, one word is live
// at a time.  PHI and J are template parameters: every index is static, the state stays in registers.
template <int PHI, int J>
ZD void keccak_xor_word(u64 st[25], const u256& w, u32 sh8) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const u32 d = w.w[7 - k];
    const int i = 8 * J + k - PHI;  // block dword that starts in this memory dword
    if (i >= 0 && i < 34) {
      const u64 le = (u64)__builtin_bswap32(sh8 ? d << sh8 : d);
      st[i >> 1] ^= (i & 1) ? le << 32 : le;
    }
    if (i - 1 >= 0 && i - 1 < 34 && sh8) {  // (wave-uniform)
      const u64 le = (u64)__builtin_bswap32(d >> (32u - sh8));
      st[(i - 1) >> 1] ^= ((i - 1) & 1) ? le << 32 : le;
    }
  }
}
// the lane's 256-bit LDS transfer slot (ZKW_XFER)
ZD u256 xfer_load(const Shared& sh, const Lane& s) {
  u256 v;
  v.w[0] = ZKW_XFER(sh, s, 0); v.w[1] = ZKW_XFER(sh, s, 1); v.w[2] = ZKW_XFER(sh, s, 2); v.w[3] = ZKW_XFER(sh, s, 3);
  v.w[4] = ZKW_XFER(sh, s, 4); v.w[5] = ZKW_XFER(sh, s, 5); v.w[6] = ZKW_XFER(sh, s, 6); v.w[7] = ZKW_XFER(sh, s, 7);
  return v;
}
ZD void xfer_store(const Shared& sh, const Lane& s, const u256& v) {
  ZKW_XFER(sh, s, 0) = v.w[0]; ZKW_XFER(sh, s, 1) = v.w[1]; ZKW_XFER(sh, s, 2) = v.w[2]; ZKW_XFER(sh, s, 3) = v.w[3];
  ZKW_XFER(sh, s, 4) = v.w[4]; ZKW_XFER(sh, s, 5) = v.w[5]; ZKW_XFER(sh, s, 6) = v.w[6]; ZKW_XFER(sh, s, 7) = v.w[7];
}
template <int J>
ZD void keccak_xor_word_phase(u64 st[25], const u256& w, u32 phase, u32 sh8) {
  switch (phase) {  // wave-uniform
    case 0: keccak_xor_word<0, J>(st, w, sh8); break;
    case 1: keccak_xor_word<1, J>(st, w, sh8); break;
    case 2: keccak_xor_word<2, J>(st, w, sh8); break;
    case 3: keccak_xor_word<3, J>(st, w, sh8); break;
    case 4: keccak_xor_word<4, J>(st, w, sh8); break;
    case 5: keccak_xor_word<5, J>(st, w, sh8); break;
    case 6: keccak_xor_word<6, J>(st, w, sh8); break;
    default: keccak_xor_word<7, J>(st, w, sh8); break;
  }
}

ZD void precompile_keccak256(ZKW_KP P, Shared& sh, Lane& s, const LogQ& q) {
  ZKW_DIV_SCOPE;  // (a lane that fails — an unreachable page, a full stream — leaves early; the others write their digest)
  const u32 in_off = q.key.w[0], in_len = q.key.w[1], out_off = q.key.w[2];
  const u32 page_r = q.key.w[4], page_w = q.key.w[5];
  u64 st[25];
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = 0;
  zk_keccak_lc_flip(st);  // the state is kept in the lane-complemented form (zk_keccak_f1600_lc)
  const u32 sh8 = (in_off & 3u) * 8u;
  u256 w_cur = u256_zero(), w_next = u256_zero();
  u32 cur_idx = 0xffffffffu, next_idx = 0xffffffffu;  // word indices held in w_cur / w_next
  const u32 n_dwords = (in_len + 3u) >> 2;
  u32 slot = 0;  // dword position inside the rate block
  u32 d_start = 0;
  // Full blocks of a wave whose calls share the byte phase of the input inside its memory word and the length — a
  // shared tape: the walk over the message is then the same for every lane, so its control flow is scalar and a block
  // is absorbed straight from the (up to five) words that hold it, with static indices.  The general loop below — one
  // stream dword at a time, word index and dword-in-word per lane, the rate block staged through a row in HBM — costs
  // four times the permutation it feeds; it keeps the last, partial block and every wave whose lanes differ.
  {
    const u32 u_phase = zkw_uniform(in_off & 31u), u_len = zkw_uniform(in_len);
    if (__ballot((in_off & 31u) != u_phase || in_len != u_len) == 0 && u_len >= ZKW_KECCAK_RATE) {
      const u32 u_sh8 = (u_phase & 3u) * 8u;
      const u32 n_full = u_len / ZKW_KECCAK_RATE;
      const u32 w0 = in_off >> 5;  // per lane
      const FatPage fpage = fat_ptr_resolve(P, sh, s, page_r);  // the page is resolved once: the call writes nothing before its reads are done
      u32 have = 0;                // words w0 .. w0 + have - 1 have been read (and witnessed), wave-uniform
#ifdef __HIP_DEVICE_COMPILE__
      const bool pf_ok = __ballot(lane_ok(s) && (!fpage.found || fpage.empty || fpage.is_aux)) == 0;  // (the requests ahead below: heap pages only, wave-uniform arena)
#endif
      for (u32 b = 0; b < n_full; b++) {
        const u32 p0 = (u_phase >> 2) + 34u * b;                    // first stream dword of the block, in dwords from word w0
        const u32 r0 = p0 >> 3, r1 = (p0 + 33u + (u_sh8 ? 1u : 0u)) >> 3;  // words the block reaches into
        // (the word a block ends in is parked in the lane's LDS transfer slot across the permutation — 8 dwords fewer to keep
        // in registers next to the 50 of the state — and taken back when the next block starts inside it)
        u256 wv = u256_zero();
#define ZKW_KECCAK_WORD(J)                                                                                       \
  {                                                                                                              \
    const u32 r = r0 + (u32)(J); /* wave-uniform */                                                              \
    if (r <= r1) {                                                                                               \
      if (r >= have) {                                                                                           \
        ZKW_DIV_IF(lane_ok(s)) {                                                                                 \
          wv = fat_page_read(sh, s, fpage, w0 + r);                                                              \
          emit_mem(P, sh, s, q.timestamp, ZKW_MEM_FAT_PTR, page_r, w0 + r, wv, false, false, 1);                 \
        }                                                                                                        \
        have = r + 1u;                                                                                           \
      } else { /* the previous block ended inside this word (only J = 0) */                                      \
        wv = xfer_load(sh, s);                                                                                   \
      }                                                                                                          \
      keccak_xor_word_phase<J>(st, wv, p0 & 7u, u_sh8);                                                          \
    }                                                                                                            \
  }
        ZKW_KECCAK_WORD(0) ZKW_KECCAK_WORD(1) ZKW_KECCAK_WORD(2) ZKW_KECCAK_WORD(3) ZKW_KECCAK_WORD(4) ZKW_KECCAK_WORD(5)
#undef ZKW_KECCAK_WORD
        xfer_store(sh, s, wv);
#ifdef __HIP_DEVICE_COMPILE__
        // the words of the NEXT block are requested before this block's permutation (two loads per word into the LDS sink:
        // no destination register, nothing to wait for — prefetch_page_words): a message word is read once and comes from
        // HBM, five dependent round trips per block otherwise, with two waves per SIMD to hide them behind
        if (pf_ok) {
          const u32 last = (u_phase + u_len - 1u) >> 5;  // last word of the message, from w0
#pragma unroll
          for (u32 j = 0; j < 5; j++)
            if (have + j <= last) prefetch_page_words(sh.heap, P.H, P.L, zkw_lds_sink_addr(), fpage.slot, fpage.hwm, (w0 + have + j) << 5);
        }
#endif
        zk_keccak_f1600_lc(st);
      }
      d_start = 34u * n_full;
      if (have) {  // the general loop continues behind the last word read
        w_cur = xfer_load(sh, s);
        cur_idx = w0 + have - 1u;
      }
    }
  }
  {
  ZKW_DIV_SCOPE;  // (the lanes' messages differ in length: a lane that is through waits behind the loop)
  for (u32 d = d_start; d < n_dwords && lane_ok(s); d++) {
    const u32 need = in_len - 4u * d < 4u ? in_len - 4u * d : 4u;  // message bytes in this stream dword
    const u32 m0 = (in_off >> 2) + d;                               // memory dword holding its first byte
    const bool two = (in_off & 3u) + need > 4u;                     // the dword straddles two memory dwords
    // fetch (in order, once) the words that hold m0 and, if used, m0 + 1
    const u32 wi0 = m0 >> 3, wi1 = (m0 + 1u) >> 3;
    ZKW_DIV_IF(wi0 != cur_idx) {
      ZKW_DIV_IF(wi0 == next_idx) {
        w_cur = w_next;
        cur_idx = next_idx;
      } else {
        w_cur = fat_ptr_read(P, sh, s, page_r, wi0);
        emit_mem(P, sh, s, q.timestamp, ZKW_MEM_FAT_PTR, page_r, wi0, w_cur, false, false, 1);
        cur_idx = wi0;
      }
    }
    u32 b = 0;
    ZKW_DIV_IF(two) {
      ZKW_DIV_IF(wi1 != cur_idx) {
        ZKW_DIV_IF(wi1 != next_idx) {
          w_next = fat_ptr_read(P, sh, s, page_r, wi1);
          emit_mem(P, sh, s, q.timestamp, ZKW_MEM_FAT_PTR, page_r, wi1, w_next, false, false, 1);
          next_idx = wi1;
        }
        b = word_be_dword(w_next, (m0 + 1u) & 7u);
      } else {
        b = word_be_dword(w_cur, (m0 + 1u) & 7u);
      }
    }
    const u32 a = word_be_dword(w_cur, m0 & 7u);
    u32 v = sh8 ? ((a << sh8) | (b >> (32u - sh8))) : a;  // big-endian stream dword
    if (need < 4u) v &= 0xffffffffu << (8u * (4u - need));
    u32 le = __builtin_bswap32(v);
    if (need < 4u) le |= 0x01u << (8u * need);  // the pad byte follows the last message byte inside this dword
    sh.krow[slot * sh.L + s.lane] = le;
    slot++;
    if (slot == ZKW_KRATE_WORDS && !(need < 4u)) {  // a full block of message bytes
      keccak_absorb_block(sh, s.lane, st);
      slot = 0;
    }
  }
  }
  if (!lane_ok(s)) return;
  // pad10*1 with the legacy 0x01 domain byte: 0x01 right after the message, 0x80 on the last byte of the block
  if ((in_len & 3u) == 0u) {
    sh.krow[slot * sh.L + s.lane] = 0x01u;
    slot++;
  }
  for (u32 p = slot; p < ZKW_KRATE_WORDS; p++) sh.krow[p * sh.L + s.lane] = 0;
  sh.krow[(ZKW_KRATE_WORDS - 1) * sh.L + s.lane] |= 0x80000000u;
  keccak_absorb_block(sh, s.lane, st);
  zk_keccak_lc_flip(st);
  u256 digest;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    digest.w[7 - 2 * i] = __builtin_bswap32((u32)st[i]);
    digest.w[6 - 2 * i] = __builtin_bswap32((u32)(st[i] >> 32));
  }
  heap_write_cur(P, sh, s, false, out_off, digest);
  emit_mem(P, sh, s, q.timestamp + 1, ZKW_MEM_HEAP, page_w, out_off, digest, false, true, 2);
}

ZD u32 zk_rotr32(u32 x, int n) { return (x >> n) | (x << (32 - n)); }

// sha256_rounds_function: `precompile_interpreted_data` rounds, two words per round from word offset
// `input_memory_offset` (the caller supplies the padded message), digest at `output_memory_offset`.
ZD void precompile_sha256(ZKW_KP P, Shared& sh, Lane& s, const LogQ& q) {
  const u32 in_word = q.key.w[0], out_off = q.key.w[2];
  const u32 page_r = q.key.w[4], page_w = q.key.w[5];
  const u32 rounds = q.key.w[6];  // low half of the u64; > 2^32 rounds cannot be paid for
  u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  u32 rd = in_word;
  ZKW_DIV_SCOPE;  // (the lanes' round counts differ)
  for (u32 round = 0; round < rounds && lane_ok(s); round++) {
    u32 w[16];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const u256 word = heap_read_cur(P, sh, s, false, rd);
      emit_mem(P, sh, s, q.timestamp, ZKW_MEM_HEAP, page_r, rd, word, false, false, 1);
      rd++;
#pragma unroll
      for (int i = 0; i < 8; i++) w[half * 8 + i] = word.w[7 - i];
    }
    u32 a = h[0], b = h[1], c = h[2], dd = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      if (i >= 16) {
        const u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
        const u32 s0 = zk_rotr32(w15, 7) ^ zk_rotr32(w15, 18) ^ (w15 >> 3);
        const u32 s1 = zk_rotr32(w2, 17) ^ zk_rotr32(w2, 19) ^ (w2 >> 10);
        w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      }
      const u32 S1 = zk_rotr32(e, 6) ^ zk_rotr32(e, 11) ^ zk_rotr32(e, 25);
      const u32 ch = (e & f) ^ (~e & g);
      const u32 t1 = hh + S1 + ch + ZKW_SHA256_K[i] + w[i & 15];
      const u32 S0 = zk_rotr32(a, 2) ^ zk_rotr32(a, 13) ^ zk_rotr32(a, 22);
      const u32 mj = (a & b) ^ (a & c) ^ (b & c);
      const u32 t2 = S0 + mj;
      hh = g; g = f; f = e; e = dd + t1; dd = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += dd; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    ZKW_DIV_IF(round == rounds - 1) {
      u256 digest;
#pragma unroll
      for (int i = 0; i < 8; i++) digest.w[7 - i] = h[i];
      heap_write_cur(P, sh, s, false, out_off, digest);
      emit_mem(P, sh, s, q.timestamp + 1, ZKW_MEM_HEAP, page_w, out_off, digest, false, true, 2);
    }
  }
}

#include "zkw_secp256k1.hip.h"

// ecrecover_function: four words from word offset `input_memory_offset` of `memory_page_to_read` (order per
// consts.ecrecover_input_layout), two words at `output_memory_offset`: the ok marker (1 / 0) and the address as the
// low 20 bytes of a big-endian word (reference test src/testing/tests/precompiles/ecrecover.rs:51-95).
ZD void precompile_ecrecover(ZKW_KP P, Shared& sh, Lane& s, const LogQ& q) {
  ZKW_DIV_SCOPE;  // (a lane with a malformed recovery id leaves early)
  const u32 in_word = q.key.w[0], out_off = q.key.w[2];
  const u32 page_r = q.key.w[4], page_w = q.key.w[5];
  u256 w0, w1, w2, w3;
  w0 = heap_read_cur(P, sh, s, false, in_word);
  emit_mem(P, sh, s, q.timestamp, ZKW_MEM_HEAP, page_r, in_word, w0, false, false, 1);
  w1 = heap_read_cur(P, sh, s, false, in_word + 1);
  emit_mem(P, sh, s, q.timestamp, ZKW_MEM_HEAP, page_r, in_word + 1, w1, false, false, 1);
  w2 = heap_read_cur(P, sh, s, false, in_word + 2);
  emit_mem(P, sh, s, q.timestamp, ZKW_MEM_HEAP, page_r, in_word + 2, w2, false, false, 1);
  w3 = heap_read_cur(P, sh, s, false, in_word + 3);
  emit_mem(P, sh, s, q.timestamp, ZKW_MEM_HEAP, page_r, in_word + 3, w3, false, false, 1);
  if (!lane_ok(s)) return;
  const bool evm_order = P.consts.ecrecover_input_layout != 0;
  const u256 vw = evm_order ? w1 : w3;
  const u256 r = evm_order ? w2 : w1;
  const u256 sg = evm_order ? w3 : w2;
  // the recovery id is a single byte that must be 0 or 1 (an assert in the precompile => the reference panics)
  if ((vw.w[0] > 1u) | ((vw.w[1] | vw.w[2] | vw.w[3] | vw.w[4] | vw.w[5] | vw.w[6] | vw.w[7]) != 0u)) {
    lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
    return;
  }
  const ec_result res = zkw_ecrecover(sh.krow + ZKW_KRATE_WORDS * sh.L + s.lane, sh.L, w0, r, sg, vw.w[0]);
  const u256 marker = u256_from_u32(res.ok);
  heap_write_cur(P, sh, s, false, out_off, marker);
  emit_mem(P, sh, s, q.timestamp + 1, ZKW_MEM_HEAP, page_w, out_off, marker, false, true, 2);
  heap_write_cur(P, sh, s, false, out_off + 1, res.address_word);
  emit_mem(P, sh, s, q.timestamp + 1, ZKW_MEM_HEAP, page_w, out_off + 1, res.address_word, false, true, 2);
}
