// BLAKE2s-256 of a batch of byte strings on gfx950 — one message per lane.
//
// The reference re-exports the `blake2` crate (src/lib.rs:21: `pub use zkevm_opcode_defs::blake2`) and never calls it:
// callers reach `Blake2s256::digest(bytes)` through `zk_evm::blake2`.  This is that function for n messages at once
// (include/zkw.h: zkw_blake2s256 / zkw_blake2s256_device), written from RFC 7693 (unkeyed, 32-byte digest, no salt or
// personalisation: parameter word 0x01010020).
//
// Layout and bound.  Message i is data[offsets[i] .. offsets[i + 1]); lane i of the grid owns it and walks its 64-byte
// blocks.  A block is fetched as the 17 aligned dwords that cover it and funnel-shifted by the byte misalignment
// (v_alignbyte), so ragged offsets cost one extra load per block, not byte loads.  The lanes of a wave read at a stride
// of one message, i.e. 64+ cache lines per block step, each of which the same lane consumes completely over the 16
// loads that follow — the lines live in L2 meanwhile; HBM sees every byte once.  The compression is 10 rounds x 8 G of
// 32-bit add / xor / rotate with the 16 message words in registers (the schedule is a compile-time permutation):
// ≈ 1.05k vector instructions per 64 bytes, so long messages are bound by the integer ALU (≈ 2.3 TB/s at 39 T
// lane-operations/s), short ones by one compression per message.  Digests leave as two 16-byte stores per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;
#ifndef ZD
#define ZD __device__ __forceinline__
#endif

namespace {
__device__ const u32 B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

ZD u32 rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }  // v_alignbit_b32

#define B2S_G(a, b, c, d, x, y) \
  a = a + b + (x);              \
  d = rotr(d ^ a, 16);          \
  c = c + d;                    \
  b = rotr(b ^ c, 12);          \
  a = a + b + (y);              \
  d = rotr(d ^ a, 8);           \
  c = c + d;                    \
  b = rotr(b ^ c, 7);
// one round with the message schedule spelled out (RFC 7693 §2.7 SIGMA row s0..s15)
#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  B2S_G(v0, v4, v8, v12, m[s0], m[s1])                                                   \
  B2S_G(v1, v5, v9, v13, m[s2], m[s3])                                                   \
  B2S_G(v2, v6, v10, v14, m[s4], m[s5])                                                  \
  B2S_G(v3, v7, v11, v15, m[s6], m[s7])                                                  \
  B2S_G(v0, v5, v10, v15, m[s8], m[s9])                                                  \
  B2S_G(v1, v6, v11, v12, m[s10], m[s11])                                                \
  B2S_G(v2, v7, v8, v13, m[s12], m[s13])                                                 \
  B2S_G(v3, v4, v9, v14, m[s14], m[s15])

// F (RFC 7693 §3.2): h <- compress(h, m, t, last)
ZD void blake2s_compress(u32 h[8], const u32 m[16], u64 t, bool last) {
  u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
  u32 v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
  u32 v12 = 0x510E527Fu ^ (u32)t, v13 = 0x9B05688Cu ^ (u32)(t >> 32), v14 = last ? ~0x1F83D9ABu : 0x1F83D9ABu, v15 = 0x5BE0CD19u;
  B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
  h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// the 16 little-endian words of the block that starts at byte `at` of `data`, bytes at or after `end` read as zero.
// `words` = the aligned dwords of the whole buffer (its allocation covers ceil(total / 4) dwords).
// LAST = false: a block that is not the message's last (end - at > 64) — all 17 covering dwords start before `end`, so
// they are loaded without a test; LAST = true: nothing at or after `end` is loaded (it may lie past the buffer).
template <bool LAST>
ZD void load_block(const u32* __restrict__ words, u64 at, u64 end, u32 m[16]) {
  const u64 w0 = at >> 2;
  const u32 sh = (u32)(at & 3u) * 8u;
  u32 w[17];
#pragma unroll
  for (int i = 0; i < 17; i++) {
    const u64 wi = w0 + (u64)i;
    w[i] = (!LAST || wi * 4u < end) ? words[wi] : 0u;
  }
  const u64 valid = end > at ? end - at : 0;  // bytes of this block that belong to the message
#pragma unroll
  for (int i = 0; i < 16; i++) {
    u32 x = sh ? (w[i] >> sh) | (w[i + 1] << (32u - sh)) : w[i];  // v_alignbyte_b32
    if (LAST) {
      const u64 lo = 4u * (u64)i;
      if (valid <= lo) x = 0;
      else if (valid < lo + 4u) x &= (1u << (8u * (u32)(valid - lo))) - 1u;
    }
    m[i] = x;
  }
}
}  // namespace

__global__ void __launch_bounds__(256) zkw_blake2s_kernel(const u32* __restrict__ words, u64 total_bytes, const u64* __restrict__ offsets, u32 n,
                                                          uint4* __restrict__ digests) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 begin = offsets[i], end = offsets[i + 1];
  (void)total_bytes;
  u32 h[8];
#pragma unroll
  for (int k = 0; k < 8; k++) h[k] = B2S_IV[k];
  h[0] ^= 0x01010020u;  // digest length 32, no key, fanout 1, depth 1
  u32 m[16];
  u64 at = begin;
  // every block but the last: the counter is the number of message bytes absorbed so far
  while (end - at > 64u) {
    load_block<false>(words, at, end, m);
    at += 64u;
    blake2s_compress(h, m, at - begin, false);
  }
  load_block<true>(words, at, end, m);  // 0..64 bytes (an empty message is one zero block), zero padded
  blake2s_compress(h, m, end - begin, true);
  digests[2 * (size_t)i] = make_uint4(h[0], h[1], h[2], h[3]);
  digests[2 * (size_t)i + 1] = make_uint4(h[4], h[5], h[6], h[7]);
}

extern "C" hipError_t zkw_launch_blake2s(const void* d_data, uint64_t total_bytes, const uint64_t* d_offsets, uint32_t n, void* d_digests,
                                         uint32_t wave_threads, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const u32 threads = wave_threads > 1 ? 256 : 1;
  hipLaunchKernelGGL(zkw_blake2s_kernel, dim3((n + threads - 1) / threads), dim3(threads), 0, stream, (const u32*)d_data, (u64)total_bytes, d_offsets, n,
                     (uint4*)d_digests);
  return hipGetLastError();
}
