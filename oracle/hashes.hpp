// ORACLE — TEST INFRASTRUCTURE ONLY (see u256.hpp header).
//
// Round functions the reference reaches through `zk_evm_abstractions::precompiles`
// (DefaultPrecompilesProcessor, call site reference src/vm_state/helpers.rs:211-213).  That
// crate (git branch v1.4.1) is not on disk; what is restated here are the published
// primitives it wraps: Keccak-f[1600] (FIPS-202 permutation; `sha3` crate's keccak::f1600)
// and the SHA-256 compression function (FIPS 180-4; `sha2` crate's compress256).
// Pinned by the reference's live tests' digests (src/testing/tests/precompiles/keccak256.rs:
// 144-196, values in SURVEY Appendix C) and by hashlib — tests/test_oracle_precompiles.py.
#pragma once
#include <cstdint>

namespace zko {

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

static inline uint64_t rotl64(uint64_t x, int n) { return n == 0 ? x : (x << n) | (x >> (64 - n)); }

inline void keccak_f1600(uint64_t st[25]) {
  static const int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int round = 0; round < 24; round++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = piln[i];
      uint64_t b = st[j];
      st[j] = rotl64(t, rotc[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KECCAK_RC[round];
  }
}

static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static const uint32_t SHA256_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

inline void sha256_compress(uint32_t state[8], const uint8_t block[64]) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++)
    w[i] = ((uint32_t)block[4 * i] << 24) | ((uint32_t)block[4 * i + 1] << 16) | ((uint32_t)block[4 * i + 2] << 8) | block[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = state[0], b = state[1], c = state[2], d = state[3], e = state[4], f = state[5], g = state[6], h = state[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + S1 + ch + SHA256_K[i] + w[i];
    uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  state[0] += a; state[1] += b; state[2] += c; state[3] += d; state[4] += e; state[5] += f; state[6] += g; state[7] += h;
}

// keccak256 of a byte string (legacy 0x01 padding), on top of the pinned permutation
inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  uint64_t st[25] = {0};
  uint8_t block[136];
  size_t off = 0;
  for (;;) {
    size_t take = len - off < 136 ? len - off : 136;
    std::memset(block, 0, sizeof block);
    std::memcpy(block, data + off, take);
    off += take;
    bool last = take < 136;
    if (last) {
      block[take] ^= 0x01;
      block[135] ^= 0x80;
    }
    for (int i = 0; i < 17; i++) {
      uint64_t w = 0;
      for (int b = 0; b < 8; b++) w |= (uint64_t)block[8 * i + b] << (8 * b);
      st[i] ^= w;
    }
    keccak_f1600(st);
    if (last) break;
  }
  for (int i = 0; i < 32; i++) out[i] = (uint8_t)(st[i / 8] >> (8 * (i % 8)));
}


// ---------------------------------------------------------------------------------------------
// BLAKE2s-256 (RFC 7693 §3, unkeyed, 32-byte digest) — the `blake2` crate the reference re-exports
// (/root/reference/src/lib.rs:21: `pub use zkevm_opcode_defs::blake2`; no call site in the reference, so the pin is
// the RFC's own test vector and hashlib.blake2s: tests/test_blake2s.py).  Written from the RFC's pseudo-code.
// ---------------------------------------------------------------------------------------------
static const uint32_t BLAKE2S_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t BLAKE2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

inline void blake2s_mix(uint32_t v[16], int a, int b, int c, int d, uint32_t x, uint32_t y) {  // RFC 7693 §3.1 (R = 16, 12, 8, 7)
  v[a] = v[a] + v[b] + x; v[d] = rotr32(v[d] ^ v[a], 16);
  v[c] = v[c] + v[d];     v[b] = rotr32(v[b] ^ v[c], 12);
  v[a] = v[a] + v[b] + y; v[d] = rotr32(v[d] ^ v[a], 8);
  v[c] = v[c] + v[d];     v[b] = rotr32(v[b] ^ v[c], 7);
}
inline void blake2s_F(uint32_t h[8], const uint8_t block[64], uint64_t t, bool last) {  // RFC 7693 §3.2
  uint32_t m[16], v[16];
  for (int i = 0; i < 16; i++) m[i] = (uint32_t)block[4 * i] | (uint32_t)block[4 * i + 1] << 8 | (uint32_t)block[4 * i + 2] << 16 | (uint32_t)block[4 * i + 3] << 24;
  for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = BLAKE2S_IV[i]; }
  v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
  if (last) v[14] = ~v[14];
  for (int r = 0; r < 10; r++) {
    const uint8_t* s = BLAKE2S_SIGMA[r];
    blake2s_mix(v, 0, 4, 8, 12, m[s[0]], m[s[1]]);   blake2s_mix(v, 1, 5, 9, 13, m[s[2]], m[s[3]]);
    blake2s_mix(v, 2, 6, 10, 14, m[s[4]], m[s[5]]);  blake2s_mix(v, 3, 7, 11, 15, m[s[6]], m[s[7]]);
    blake2s_mix(v, 0, 5, 10, 15, m[s[8]], m[s[9]]);  blake2s_mix(v, 1, 6, 11, 12, m[s[10]], m[s[11]]);
    blake2s_mix(v, 2, 7, 8, 13, m[s[12]], m[s[13]]); blake2s_mix(v, 3, 4, 9, 14, m[s[14]], m[s[15]]);
  }
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
inline void blake2s256(const uint8_t* data, size_t len, uint8_t out[32]) {  // RFC 7693 §3.3, kk = 0, nn = 32
  uint32_t h[8];
  for (int i = 0; i < 8; i++) h[i] = BLAKE2S_IV[i];
  h[0] ^= 0x01010000u ^ 32u;
  uint8_t block[64];
  size_t off = 0;
  while (len - off > 64) {
    blake2s_F(h, data + off, (uint64_t)off + 64, false);
    off += 64;
  }
  std::memset(block, 0, sizeof block);
  if (len > off) std::memcpy(block, data + off, len - off);
  blake2s_F(h, block, (uint64_t)len, true);
  for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)h[i]; out[4 * i + 1] = (uint8_t)(h[i] >> 8); out[4 * i + 2] = (uint8_t)(h[i] >> 16); out[4 * i + 3] = (uint8_t)(h[i] >> 24); }
}

}  // namespace zko
