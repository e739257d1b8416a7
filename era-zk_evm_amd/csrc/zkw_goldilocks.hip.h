// Goldilocks field arithmetic and the "ZKW-GL-sponge v2" permutation / leaf / chain step (the build's own spec, see
// zkw_commit.hip and DESIGN.md §commitments).  Shared by the commitment kernels (zkw_commit.hip) and by the cycle kernel,
// which chains the decommit queue while it runs (zkw_kernels.hip: op_far_call).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef ZD
#define ZD __device__ __forceinline__
#endif
// The round constants are read through the constant address space: uniform addresses, so the loads are scalar
// (s_load_dwordx8/x16 through the scalar cache, issued ahead by the compiler) instead of a vector-memory round trip per
// constant — a lone wave chaining inside the cycle kernel otherwise waits for 118 dependent global loads per permutation.
#ifdef __HIP_DEVICE_COMPILE__
#define ZKW_GL_RC const u64 __attribute__((address_space(4)))*
#else
#define ZKW_GL_RC const u64*
#endif
typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------
// Goldilocks
// ---------------------------------------------------------------------------------------------
#define GL_P 0xffffffff00000001ULL
#define GL_EPS 0xffffffffULL

ZD u64 gl_add(u64 a, u64 b) {
  u64 r = a + b;
  if (r < a) r += GL_EPS;  // wrapped past 2^64: 2^64 = EPS (mod p); cannot wrap again since a, b < p
  if (r >= GL_P) r -= GL_P;
  return r;
}
ZD void mul64(u64 a, u64 b, u64& lo, u64& hi) {
  const u64 a0 = (u32)a, a1 = a >> 32, b0 = (u32)b, b1 = b >> 32;
  const u64 p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  const u64 mid = (p00 >> 32) + (u32)p01 + (u32)p10;
  lo = (mid << 32) | (u32)p00;
  hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}
ZD u64 gl_reduce128(u64 lo, u64 hi) {
  const u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;  // borrow: subtract 2^64 = EPS (mod p) once more
  const u64 t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1) without a multiply (integer multiplies are quarter rate)
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;
  if (r >= GL_P) r -= GL_P;
  return r;
}
ZD u64 gl_mul(u64 a, u64 b) {
  u64 lo, hi;
  mul64(a, b, lo, hi);
  return gl_reduce128(lo, hi);
}
// a^2: the two cross products are equal (3 wide multiplies instead of 4)
ZD void sqr64(u64 a, u64& lo, u64& hi) {
  const u64 a0 = (u32)a, a1 = a >> 32;
  const u64 p00 = a0 * a0, p01 = a0 * a1, p11 = a1 * a1;
  const u64 mid = (p00 >> 32) + 2 * (u64)(u32)p01;
  lo = (mid << 32) | (u32)p00;
  hi = p11 + 2 * (p01 >> 32) + (mid >> 32);
}
// reduction without the final conditional subtraction: the result is < 2^64 and congruent, possibly >= p.  Such a value
// is a valid INPUT of mul64 / sqr64 / gl_reduce128 (they accept any u64), so the canonical form is only restored at the
// end of a multiplication chain.
ZD u64 gl_reduce128_lazy(u64 lo, u64 hi) {
  const u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;
  const u64 t1 = (hi_lo << 32) - hi_lo;
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;
  return r;
}
// a * b, reduced lazily (any u64 in, a congruent u64 out) — the S-box is four of these.  Written out in gfx950
// instructions: v_mad_u64_u32 is a full-rate 32 x 32 + 64 multiply-add WITH a carry out, which the compiler never uses;
// with it the product and the fold 2^64 = 2^32 - 1, 2^96 = -1 (mod p) take 19 instructions instead of ~30:
//   a b = p00 + 2^32 (p01 + p10) + 2^64 p11,  M = p01 + p10 = m0 + 2^32 m1 + 2^64 cm
//       = [p00 + 2^32 m0]  +  (2^32 - 1) (m1 + p11.lo + c1)  -  (p11.hi + c2 + cm)        (c1, c2: the carries of the sums)
//       =  X               +  EPS q                           -  h
// T = X + EPS q is one multiply-add (carry c3), U = T - h one subtraction (borrow b); the result is U + (c3 - b) EPS,
// which neither wraps nor goes negative (|value| bounds in profiles/tools/mulred_check.py, which also checks the sequence
// against integer arithmetic).  A scalar register written by a vector instruction (vcc or a carry pair) may be read by a
// vector instruction two issue slots later at the earliest on gfx940+: the s_nops, where no independent instruction
// fits.  The temporaries are fixed registers (an asm operand cannot name the halves of a 64-bit register pair).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKW_GL_PORTABLE)
#ifdef ZKW_GL_TIMING_NO_NOPS  /* profiles/tools/perm_probe.hip only: what the wait states cost (results are then wrong) */
#define ZKW_GL_NOP1
#define ZKW_GL_NOP0
#else
#define ZKW_GL_NOP1 "s_nop 1\n\t"
#define ZKW_GL_NOP0 "s_nop 0\n\t"
#endif
ZD u64 gl_mulred(u64 a, u64 b) {
  u32 r0, r1;
  u64 s0, s1;
  asm("v_mad_u64_u32 v[112:113], %2, %4, %6, 0\n\t"             // P = a0 b0
      "v_mad_u64_u32 v[114:115], %2, %4, %7, 0\n\t"             // M = a0 b1
      "v_mad_u64_u32 v[116:117], %3, %5, %7, 0\n\t"             // Q = a1 b1
      "v_mad_u64_u32 v[114:115], %2, %5, %6, v[114:115]\n\t"    // M += a1 b0, cm -> %2
      "v_add_co_u32 v113, vcc, v113, v114\n\t"                  // X.hi = P.hi + m0, c1
      ZKW_GL_NOP1
      "v_addc_co_u32 v115, vcc, v115, v116, vcc\n\t"            // q = m1 + Q.lo + c1, c2
      "v_mad_u64_u32 v[112:113], %3, v115, -1, v[112:113]\n\t"  // T = X + EPS q, c3 -> %3
      ZKW_GL_NOP0
      "v_addc_co_u32 v117, vcc, 0, v117, vcc\n\t"               // h = Q.hi + c2
      "v_cndmask_b32 v118, 0, 1, %3\n\t"                        // e = c3
      "v_addc_co_u32 v117, %3, 0, v117, %2\n\t"                 // h += cm
      "v_sub_co_u32 v112, vcc, v112, v117\n\t"                  // U = T - h
      ZKW_GL_NOP1
      "v_subbrev_co_u32 v113, vcc, 0, v113, vcc\n\t"            // borrow b
      ZKW_GL_NOP1
      "v_subbrev_co_u32 v118, vcc, 0, v118, vcc\n\t"            // e = c3 - b
      "v_mad_i64_i32 v[112:113], %2, v118, -1, v[112:113]\n\t"  // U - e
      "v_add_u32 %1, v113, v118\n\t"                            // + e 2^32
      "v_mov_b32 %0, v112"
      : "=&v"(r0), "=&v"(r1), "=&s"(s0), "=&s"(s1)
      : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
      : "vcc", "v112", "v113", "v114", "v115", "v116", "v117", "v118");
  return ((u64)r1 << 32) | r0;
}
#else
ZD u64 gl_mulred(u64 a, u64 b) {
  u64 lo, hi;
  mul64(a, b, lo, hi);
  return gl_reduce128_lazy(lo, hi);
}
#endif
ZD u64 gl_pow7(u64 x) {
  const u64 x2 = gl_mulred(x, x);
  const u64 x3 = gl_mulred(x2, x);
  const u64 x4 = gl_mulred(x2, x2);
  return gl_mulred(x4, x3);
}

// The linear layers work on the 32-bit halves of the elements.  A 128-bit accumulator costs a carry chain per addition,
// and on gfx940+ every link of a carry chain (v_add_co -> v_addc_co) is followed by two wait states (a scalar register
// written by a vector instruction): the 128-bit form of the layers was one third s_nop.  Instead the low words and the
// high words of the state go through the (integer) matrix separately — the coefficients are small, so each half stays
// far below 2^64 and every step is one carry-free v_mad_u64_u32 (32 x 32 + 64) or v_lshl_add_u64 — and the two halves
// L, H of an output are folded once:  L + 2^32 H  (mod p), with 2^64 = 2^32 - 1.
// Inside the permutation every value is kept only congruent (< 2^64, possibly >= p): products, the sums below and the
// round-constant addition accept that, and gl_permute canonicalises the state once at the end.
//
// a * b + c for operands whose result fits 64 bits (the carry out of the instruction goes to a scratch scalar pair)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKW_GL_PORTABLE)
template <int K>
ZD u64 gl_madk(u32 a, u64 c) {  // a * K + c, K an inline constant (-16..64; -1 is 2^32 - 1 as a 32-bit operand)
  static_assert(K >= -16 && K <= 64, "inline constant");
  u64 d, sc;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(sc) : "v"(a), "n"(K), "v"(c));
  return d;
}
template <int K>
ZD u64 gl_mulk(u32 a) {  // a * K as 64 bits (also the zero extension, K = 1, without a second register to clear)
  static_assert(K >= 0 && K <= 64, "inline constant");
  u64 d, sc;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(sc) : "v"(a), "n"(K));
  return d;
}
ZD u64 gl_mads(u32 a, u32 k, u64 c) {  // a * k + c, k uniform (a scalar register: VOP3 takes no literal on gfx9)
  u64 d, sc;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(sc) : "v"(a), "s"(k), "v"(c));
  return d;
}
#else
template <int K>
ZD u64 gl_madk(u32 a, u64 c) { return (u64)a * (u32)K + c; }
template <int K>
ZD u64 gl_mulk(u32 a) { return (u64)a * (u32)K; }
ZD u64 gl_mads(u32 a, u32 k, u64 c) { return (u64)a * k + c; }
#endif
// L + 2^32 H (mod p) for L < 2^63 and H < 2^63 with (H >> 32) (2^32 - 1) + L < 2^64 (here L, H < 2^45): the high word of
// H folds in through 2^64 = 2^32 - 1 without a carry; adding the low word of H at bit 32 can wrap once, and the wrapped
// value is below 2^46, so the correction 2^64 = 2^32 - 1 cannot wrap again.  Any u64 out, congruent.
ZD u64 gl_fold_halves(u64 L, u64 H) {
  const u64 c = gl_madk<-1>((u32)(H >> 32), L);
  const u32 hi = (u32)(c >> 32) + (u32)H;
  const u32 e = hi < (u32)H ? 0xffffffffu : 0u;
  return gl_madk<1>(e, ((u64)hi << 32) | (u32)c);
}
// s + rc for any s < 2^64 and a canonical constant: s + rc < 2^65 - 2^32, so one wrap correction suffices
ZD u64 gl_add_rc(u64 s, u64 rc) {
  u64 x = s + rc;
  if (x < s) x += GL_EPS;
  return x;
}

// M4 of the Poseidon2 paper: [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] on 32-bit words, exact (row sums <= 16, so the
// outputs are below 2^36)
ZD void gl_m4_words(u32 a, u32 b, u32 c, u32 d, u64& o0, u64& o1, u64& o2, u64& o3) {
  const u64 t0 = gl_madk<1>(a, gl_mulk<1>(b)), t1 = gl_madk<1>(c, gl_mulk<1>(d));
  const u64 t2 = gl_madk<2>(b, t1), t3 = gl_madk<2>(d, t0);
  const u64 t4 = (t1 << 2) + t3, t5 = (t0 << 2) + t2;
  o0 = t3 + t5; o1 = t5; o2 = t2 + t4; o3 = t4;
}
// external layer circ(2 M4, M4, M4): each half below 2^38
ZD void gl_external(u64 s[12]) {
  u64 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i += 4) {
    gl_m4_words((u32)s[i], (u32)s[i + 1], (u32)s[i + 2], (u32)s[i + 3], lo[i], lo[i + 1], lo[i + 2], lo[i + 3]);
    gl_m4_words((u32)(s[i] >> 32), (u32)(s[i + 1] >> 32), (u32)(s[i + 2] >> 32), (u32)(s[i + 3] >> 32), hi[i], hi[i + 1],
                hi[i + 2], hi[i + 3]);
  }
  u64 sl[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    sl[j] = lo[j] + lo[4 + j] + lo[8 + j];
    sh[j] = hi[j] + hi[4 + j] + hi[8 + j];
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_fold_halves(lo[i] + sl[i & 3], hi[i] + sh[i & 3]);
}
// internal layer  s_i' = sum + 2^i s_i: the word sums are below 2^36, a half of an output below 2^44
ZD void gl_internal(u64 s[12]) {
  u64 sl = gl_mulk<1>((u32)s[0]), sh = gl_mulk<1>((u32)(s[0] >> 32));
#pragma unroll
  for (int i = 1; i < 12; i++) {
    sl = gl_madk<1>((u32)s[i], sl);
    sh = gl_madk<1>((u32)(s[i] >> 32), sh);
  }
#define GL_INT_K(i) s[i] = gl_fold_halves(gl_madk<(1 << i)>((u32)s[i], sl), gl_madk<(1 << i)>((u32)(s[i] >> 32), sh))
#define GL_INT_S(i) s[i] = gl_fold_halves(gl_mads((u32)s[i], 1u << i, sl), gl_mads((u32)(s[i] >> 32), 1u << i, sh))
  GL_INT_K(0); GL_INT_K(1); GL_INT_K(2); GL_INT_K(3); GL_INT_K(4); GL_INT_K(5); GL_INT_K(6);
  GL_INT_S(7); GL_INT_S(8); GL_INT_S(9); GL_INT_S(10); GL_INT_S(11);
#undef GL_INT_K
#undef GL_INT_S
}

ZD void gl_permute(const u64* rc_, u64 s[12]) {
  ZKW_GL_RC rc = (ZKW_GL_RC)rc_;
  gl_external(s);
  int k = 0;
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_pow7(gl_add_rc(s[i], rc[k + i]));
    k += 12;
    gl_external(s);
  }
  for (int r = 0; r < 22; r++) {
    s[0] = gl_pow7(gl_add_rc(s[0], rc[k]));
    k += 1;
    gl_internal(s);
  }
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_pow7(gl_add_rc(s[i], rc[k + i]));
    k += 12;
    gl_external(s);
  }
#pragma unroll
  for (int i = 0; i < 12; i++)
    if (s[i] >= GL_P) s[i] -= GL_P;  // canonical representatives out
}

// sponge over n <= 32 field elements held in a statically indexed array
template <int N>
ZD void gl_leaf(const u64* rc, u32 type, const u64 f[N], u64 out[4]) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = 0;
  s[8] = ((u64)type << 32) | (u64)N;
#pragma unroll
  for (int b = 0; b < (N + 7) / 8; b++) {
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (b * 8 + j < N) s[j] = gl_add(s[j], f[b * 8 + j]);
    gl_permute(rc, s);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = s[i];
}

// tail' = P(leaf | tail | index + 1 | queue | x10 | x11)[0..4].  x10 / x11 are zero except for the decommit queue, whose
// per-record fields ride there (timestamp | fresh << 32, page) next to a leaf that depends only on the code (cached per
// preimage at upload): one permutation per decommit.
ZD void gl_chain_step(const u64* rc, const u64 leaf[4], u64 tail[4], u64 index_plus_1, u32 queue_id, u64 x10 = 0, u64 x11 = 0) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s[i] = leaf[i];
    s[4 + i] = tail[i];
  }
  s[8] = index_plus_1;
  s[9] = queue_id;
  s[10] = x10;
  s[11] = x11;
  gl_permute(rc, s);
#pragma unroll
  for (int i = 0; i < 4; i++) tail[i] = s[i];
}

// Memory / log queues: the record is the input of the chain permutation (no leaf).  `w` = the record's 2K u32 words;
// elements below 2^56: K pairs (even word | low 24 bits of the odd word << 32), then the dropped top bytes, seven per
// element; every block of 7 elements is one permutation  tail' = P(e | tail | (j + 1) | queue << 40 | block << 48)[0..4].
template <int K>
ZD void gl_chain_record(const u64* rc, const u32 w[2 * K], u64 tail[4], u64 index_plus_1, u32 queue_id) {
  constexpr int NE = K + (K + 6) / 7;  // elements
  u64 e[NE];
#pragma unroll
  for (int i = 0; i < K; i++) e[i] = (u64)w[2 * i] | ((u64)(w[2 * i + 1] & 0xffffffu) << 32);
#pragma unroll
  for (int f = 0; f < (K + 6) / 7; f++) {
    u64 x = 0;
#pragma unroll
    for (int i = 0; i < 7; i++)
      if (7 * f + i < K) x |= (u64)(w[2 * (7 * f + i) + 1] >> 24) << (8 * i);
    e[K + f] = x;
  }
#pragma unroll 1
  for (int b = 0; b < (NE + 6) / 7; b++) {
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 7; i++) {
      u64 v = 0;
#pragma unroll
      for (int k = 0; k < NE; k++)
        if (k == 7 * b + i) v = e[k];  // static indexing: `e` stays in registers
      s[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) s[7 + i] = tail[i];
    s[11] = index_plus_1 | ((u64)queue_id << 40) | ((u64)b << 48);
    gl_permute(rc, s);
#pragma unroll
    for (int i = 0; i < 4; i++) tail[i] = s[i];
  }
}
