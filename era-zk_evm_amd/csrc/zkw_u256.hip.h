// 256-bit integer arithmetic for gfx950 lanes (one VM instance per lane).
//
// Values are 8 x u32 little-endian limbs held in VGPRs; every loop below has compile-time
// bounds and static indices so nothing is demoted to scratch.  Semantics are those of
// ethereum_types::U256 as used by the reference ALU (reference src/opcodes/execution/
// add.rs:35, sub.rs:35, mul.rs:35-39, div.rs:50, shift.rs:44-62, uma.rs:299-361).
//
// Division is a fixed-trip-count Knuth algorithm D: the divisor is normalised to a full
// 256 bits (so every lane of a wave runs the same 9 digit steps — no divergence on operand
// length), each quotient digit comes from a 2-by-1 division with a precomputed reciprocal
// (Möller & Granlund, "Improved division by invariant integers", Alg. 4) followed by the
// classical two-limb correction and a conditional add-back.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

struct u256 {
  u32 w[8];
};

#define ZD __device__ __forceinline__

ZD u256 u256_zero() {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = 0;
  return r;
}
ZD u256 u256_from_u32(u32 v) {
  u256 r = u256_zero();
  r.w[0] = v;
  return r;
}
ZD bool u256_is_zero(const u256& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3] | a.w[4] | a.w[5] | a.w[6] | a.w[7]) == 0; }
ZD bool u256_eq(const u256& a, const u256& b) {
  u32 d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d |= a.w[i] ^ b.w[i];
  return d == 0;
}
ZD u256 u256_from_uint4(uint4 lo, uint4 hi) {
  u256 r;
  r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
  r.w[4] = hi.x; r.w[5] = hi.y; r.w[6] = hi.z; r.w[7] = hi.w;
  return r;
}
ZD uint4 u256_lo4(const u256& a) { return make_uint4(a.w[0], a.w[1], a.w[2], a.w[3]); }
ZD uint4 u256_hi4(const u256& a) { return make_uint4(a.w[4], a.w[5], a.w[6], a.w[7]); }

// overflowing_add (add.rs:35)
ZD u256 u256_add(const u256& a, const u256& b, bool& of) {
  u256 r;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (u64)a.w[i] + b.w[i];
    r.w[i] = (u32)c;
    c >>= 32;
  }
  of = c != 0;
  return r;
}
// overflowing_sub (sub.rs:35)
ZD u256 u256_sub(const u256& a, const u256& b, bool& of) {
  u256 r;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 d = (u64)a.w[i] - b.w[i] - borrow;
    r.w[i] = (u32)d;
    borrow = (d >> 32) & 1;
  }
  of = borrow != 0;
  return r;
}
// full_mul (mul.rs:35-39): 64 x v_mad_u64_u32
ZD void u256_mul(const u256& a, const u256& b, u256& lo, u256& hi) {
  u32 out[16];
#pragma unroll
  for (int i = 0; i < 16; i++) out[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      u64 t = (u64)a.w[i] * b.w[j] + out[i + j] + carry;
      out[i + j] = (u32)t;
      carry = (u32)(t >> 32);
    }
    out[i + 8] = carry;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    lo.w[i] = out[i];
    hi.w[i] = out[i + 8];
  }
}

// U256 << n with the `uint` crate's "n >= 256 => 0" (shift.rs:51,58; uma.rs:303,361)
// funnel shift: low 32 bits of ((hi:lo) >> (n & 31)) — one v_alignbit_b32.  (Spelling the funnel as
// ((u64)hi << 32 | lo) lets the optimiser fuse the two limb loads into an unaligned i64 load while the
// operand still sits in a struct, which then pins the whole operand in scratch memory.)
ZD u32 zk_funnel_r(u32 hi, u32 lo, u32 n) { return __builtin_amdgcn_alignbit(hi, lo, n); }

// Per-lane selects are bit selects with an all-ones / all-zero mask — (m & a) | (~m & b), one v_bfi_b32 and no lane
// mask in a scalar register.  Written as `if (ws & 1) { move all limbs }`, or as `cond ? a : b` per limb (which the
// optimiser turns back into a branch when many selects share a condition), a word-shift stage is a full copy of the
// value plus an exec-masked block of eight moves: 72 instructions per shift instead of ~40.
ZD u32 zk_sel(u32 m, u32 a, u32 b) { return (m & a) | (~m & b); }
ZD u32 zk_mask(bool c) { return c ? 0xffffffffu : 0u; }

ZD u256 u256_shl(const u256& a, u32 n) {
  const u32 bs = n & 31;
  const u32 m1 = zk_mask((n & 32u) != 0), m2 = zk_mask((n & 64u) != 0), m4 = zk_mask((n & 128u) != 0), nz = zk_mask(n < 256);
  const u32 mb = zk_mask(bs != 0);
  u32 t[8], u[8], v[8];
  t[0] = a.w[0] << bs;
#pragma unroll
  for (int i = 1; i < 8; i++) t[i] = zk_sel(mb, zk_funnel_r(a.w[i], a.w[i - 1], 32u - bs), a.w[i]);
#pragma unroll
  for (int i = 0; i < 8; i++) u[i] = zk_sel(m1, i >= 1 ? t[i >= 1 ? i - 1 : 0] : 0u, t[i]);
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = zk_sel(m2, i >= 2 ? u[i >= 2 ? i - 2 : 0] : 0u, u[i]);
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = zk_sel(m4, i >= 4 ? v[i >= 4 ? i - 4 : 0] : 0u, v[i]) & nz;
  return r;
}
ZD u256 u256_shr(const u256& a, u32 n) {
  const u32 bs = n & 31;
  const u32 m1 = zk_mask((n & 32u) != 0), m2 = zk_mask((n & 64u) != 0), m4 = zk_mask((n & 128u) != 0), nz = zk_mask(n < 256);
  u32 t[8], u[8], v[8];
#pragma unroll
  for (int i = 0; i < 7; i++) t[i] = zk_funnel_r(a.w[i + 1], a.w[i], bs);
  t[7] = a.w[7] >> bs;
#pragma unroll
  for (int i = 0; i < 8; i++) u[i] = zk_sel(m1, i <= 6 ? t[i <= 6 ? i + 1 : 7] : 0u, t[i]);
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = zk_sel(m2, i <= 5 ? u[i <= 5 ? i + 2 : 7] : 0u, u[i]);
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = zk_sel(m4, i <= 3 ? v[i <= 3 ? i + 4 : 7] : 0u, v[i]) & nz;
  return r;
}
// the low `nbits` bits set (0 <= nbits <= 256): all-ones shifted right by 256 - nbits, limb i a funnel over the
// extended all-ones value
ZD u256 u256_low_mask(u32 nbits) {
  const u32 t = 256u - nbits, q = t >> 5, r = t & 31u;
  u32 o[9];
#pragma unroll
  for (int j = 0; j < 9; j++) o[j] = zk_mask(q + (u32)j <= 7u);  // limb j + q of the extended value
  u256 m;
#pragma unroll
  for (int i = 0; i < 8; i++) m.w[i] = zk_funnel_r(o[i + 1], o[i], r);
  return m;
}
// (m & a) | (~m & b) per limb — v_bfi_b32
ZD u256 u256_select_bits(const u256& m, const u256& a, const u256& b) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = zk_sel(m.w[i], a.w[i], b.w[i]);
  return r;
}
// the 32 bytes that start `unal` bytes (0..31) into the 64-byte big-endian string hi ‖ lo: (hi << 8 unal) | (lo >> (256 - 8 unal))
// as ONE window over the 16 limbs instead of two shifts and an or (uma.rs:291-300)
ZD u256 u256_byte_window(const u256& hi, const u256& lo, u32 unal) {
  const u32 q = unal >> 2, r = (unal & 3u) * 8u, p = 7u - q;
  const u32 m4 = zk_mask((p & 4u) != 0), m2 = zk_mask((p & 2u) != 0), m1 = zk_mask((p & 1u) != 0), mr = zk_mask(r != 0);
  u32 c[16], f[12], g[10], e[9];
#pragma unroll
  for (int i = 0; i < 8; i++) { c[i] = lo.w[i]; c[i + 8] = hi.w[i]; }
#pragma unroll
  for (int j = 0; j < 12; j++) f[j] = zk_sel(m4, c[j + 4], c[j]);
#pragma unroll
  for (int j = 0; j < 10; j++) g[j] = zk_sel(m2, f[j + 2], f[j]);
#pragma unroll
  for (int j = 0; j < 9; j++) e[j] = zk_sel(m1, g[j + 1], g[j]);   // e[j] = c[j + 7 - q]
  u256 out;
#pragma unroll
  for (int i = 0; i < 8; i++) out.w[i] = zk_sel(mr, zk_funnel_r(e[i + 1], e[i], 32u - r), e[i + 1]);
  return out;
}
// The same window, and the merge of a written value into the two words it straddles, for a byte offset that is the SAME for
// every lane of the wave (a shared tape with an immediate or a common cursor): the dword part Q of the offset is a template
// parameter (the caller switches on it: a scalar branch), the byte part `b8` = 8 * (offset & 3) a scalar — every index is
// static and a dword costs one funnel shift: ~10 instructions instead of ~50 (window) / ~270 (the six per-lane 256-bit
// shifts of the merge).  offset = 4 Q + b8 / 8, 0 < offset < 32.
template <int Q>
ZD u256 u256_byte_window_at(const u256& hi, const u256& lo, u32 b8) {
  u32 c[16];
#pragma unroll
  for (int i = 0; i < 8; i++) { c[i] = lo.w[i]; c[i + 8] = hi.w[i]; }
  u256 out;
#pragma unroll
  for (int i = 0; i < 8; i++) out.w[i] = b8 ? zk_funnel_r(c[i + 8 - Q], c[i + 7 - Q], 32u - b8) : c[i + 8 - Q];
  return out;
}
// (w0 : w1) with the 32 bytes at byte `offset` of w0 replaced by `v` -> (n0 : n1); big-endian bytes, w[7] most significant
template <int Q>
ZD void u256_merge_at(const u256& w0, const u256& w1, const u256& v, u32 b8, u256& n0, u256& n1) {
  u32 t[16];
#pragma unroll
  for (int i = 0; i < 8; i++) { t[i] = w1.w[i]; t[i + 8] = w0.w[i]; }
  if (b8 == 0) {  // (wave-uniform) the value covers dwords 8 - Q .. 15 - Q
#pragma unroll
    for (int i = 0; i < 8; i++) t[8 - Q + i] = v.w[i];
  } else {  // the value starts 32 - b8 bits into dword D = 7 - Q and ends 32 - b8 bits into dword D + 8
    constexpr int D = 7 - Q;
    const u32 keep_lo = 0xffffffffu >> b8;
    t[D] = (t[D] & keep_lo) | (v.w[0] << (32u - b8));
#pragma unroll
    for (int i = 1; i < 8; i++) t[D + i] = zk_funnel_r(v.w[i], v.w[i - 1], b8);
    t[D + 8] = (t[D + 8] & ~keep_lo) | (v.w[7] >> b8);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) { n1.w[i] = t[i]; n0.w[i] = t[i + 8]; }
}
ZD u256 u256_or(const u256& a, const u256& b) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = a.w[i] | b.w[i];
  return r;
}
ZD u256 u256_and(const u256& a, const u256& b) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = a.w[i] & b.w[i];
  return r;
}
ZD u256 u256_xor(const u256& a, const u256& b) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = a.w[i] ^ b.w[i];
  return r;
}

// count leading zero bits of a non-zero u256
ZD u32 u256_clz(const u256& a) {
  u32 n = 0;
  bool done = false;
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    u32 z = a.w[i] ? (u32)__builtin_clz(a.w[i]) : 32u;
    n += done ? 0u : z;
    done = done || (a.w[i] != 0);
  }
  return n;
}

// floor((2^64 - 1) / d) - 2^32 for a normalised d (top bit set): the 2-by-1 reciprocal
ZD u32 zk_invert_limb(u32 d) { return (u32)(0xffffffffffffffffULL / d - 0x100000000ULL); }

// (u1:u0) / d with u1 < d, d normalised, v = zk_invert_limb(d) — Möller–Granlund Alg. 4
ZD u32 zk_div_2by1(u32 u1, u32 u0, u32 d, u32 v, u32& rem) {
  u64 q = (u64)v * u1 + (((u64)u1 << 32) | u0);
  u32 q1 = (u32)(q >> 32) + 1;
  u32 q0 = (u32)q;
  u32 r = u0 - q1 * d;
  if (r > q0) {
    q1 -= 1;
    r += d;
  }
  if (r >= d) {
    q1 += 1;
    r -= d;
  }
  rem = r;
  return q1;
}

// div_mod (div.rs:50); b != 0
ZD void u256_divmod(const u256& a, const u256& b, u256& q, u256& r) {
  const u32 s = u256_clz(b);  // 0..255
  const u256 v = u256_shl(b, s);
  // u = a << s as 16 limbs + overflow limb (always 0 since a < 2^256 and s < 256 => u < 2^512)
  const u256 ulo = u256_shl(a, s);
  const u256 uhi = s ? u256_shr(a, 256 - s) : u256_zero();
  const u32 dinv = zk_invert_limb(v.w[7]);
  // window rem[0..8]: rem[8] is the top limb; starts as (0 : uhi), then slides down over ulo
  u32 rem[9];
#pragma unroll
  for (int i = 0; i < 8; i++) rem[i] = uhi.w[i];
  rem[8] = 0;
  // uhi < 2^s <= v, so the Knuth invariant (window / v < B) holds from the start and the
  // quotient has exactly 8 digits: steps j = 7..0, each sliding one limb of ulo into the window.
#pragma unroll
  for (int j = 7; j >= 0; j--) {
#pragma unroll
    for (int i = 8; i >= 1; i--) rem[i] = rem[i - 1];
    rem[0] = ulo.w[j];
#if defined(__HIP_DEVICE_COMPILE__) || (defined(ZKW_EMU_WAVE) && ZKW_EMU_WAVE > 1) /* (real waves: the device, or the 64-lane emulation of tests/emu) */
    // a step whose window is below the divisor in every lane of the wave (top limb of the window zero, the next one below the
    // divisor's top limb) produces the digit 0 and leaves the window as it is: skipped as a scalar branch.  Operands of similar
    // size — the common case — need one or two of the eight steps.
#ifdef __HIP_DEVICE_COMPILE__
    if (__builtin_amdgcn_ballot_w64((rem[8] != 0) | (rem[7] >= v.w[7])) == 0) {
#else
    if (__ballot((rem[8] != 0) | (rem[7] >= v.w[7])) == 0) {
#endif
      q.w[j] = 0;
      continue;
    }
#endif
    // estimate qhat from (rem[8]:rem[7]) / v[7]
    u32 qhat, rhat;
    bool rhat_big = false;  // rhat >= B: no further correction possible/needed
    if (rem[8] >= v.w[7]) {
      qhat = 0xffffffffu;
      u64 rh = (u64)rem[7] + v.w[7];  // rhat = rem[8]*B + rem[7] - qhat*v7 with rem[8] == v7
      rhat = (u32)rh;
      rhat_big = (rh >> 32) != 0;
    } else {
      qhat = zk_div_2by1(rem[8], rem[7], v.w[7], dinv, rhat);
    }
#pragma unroll
    for (int it = 0; it < 2; it++) {
      u64 lhs = (u64)qhat * v.w[6];
      u64 rhs = ((u64)rhat << 32) | rem[6];
      if (!rhat_big && lhs > rhs) {
        qhat -= 1;
        u64 rh = (u64)rhat + v.w[7];
        rhat = (u32)rh;
        rhat_big = (rh >> 32) != 0;
      }
    }
    // multiply and subtract: rem -= qhat * v
    u32 carry = 0;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      u64 p = (u64)qhat * v.w[i] + carry;
      carry = (u32)(p >> 32);
      u64 t = (u64)rem[i] - (u32)p - borrow;
      rem[i] = (u32)t;
      borrow = (u32)(t >> 32) & 1;
    }
    u64 t = (u64)rem[8] - carry - borrow;
    rem[8] = (u32)t;
    if ((t >> 32) & 1) {  // went negative: add back once
      qhat -= 1;
      u64 c = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        c += (u64)rem[i] + v.w[i];
        rem[i] = (u32)c;
        c >>= 32;
      }
      rem[8] += (u32)c;
    }
    q.w[j] = qhat;
  }
  // remainder = rem >> s
  u256 rn;
#pragma unroll
  for (int i = 0; i < 8; i++) rn.w[i] = rem[i];
  r = u256_shr(rn, s);
}
