// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load it.
//
// 256-bit unsigned integer with the semantics of `ethereum_types::U256` (the `uint` crate)
// as used by the reference hot path: 4 x u64 little-endian limbs (`.0[0]` lowest, reference
// src/opcodes/execution/mul.rs:36-39), overflowing_add/sub (add.rs:35, sub.rs:35), full_mul
// (mul.rs:35), div_mod (div.rs:50), shl/shr with "shift >= 256 => 0" (shift.rs:48-62,
// uma.rs:299-303).  The crate itself is a third-party dependency that is not on disk
// (SURVEY.md fact 4); tests/test_oracle_u256.py pins these routines against Python integers.
#pragma once
#include <cstdint>
#include <cstring>

namespace zko {

struct U256 {
  uint64_t l[4];
  static U256 zero() { return U256{{0, 0, 0, 0}}; }
  static U256 from_u64(uint64_t v) { return U256{{v, 0, 0, 0}}; }
  static U256 from_u128(uint64_t lo, uint64_t hi) { return U256{{lo, hi, 0, 0}}; }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  uint64_t low_u64() const { return l[0]; }
  uint32_t low_u32() const { return (uint32_t)l[0]; }
  bool operator==(const U256& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
  bool operator!=(const U256& o) const { return !(*this == o); }
};

inline int cmp(const U256& a, const U256& b) {
  for (int i = 3; i >= 0; i--) {
    if (a.l[i] < b.l[i]) return -1;
    if (a.l[i] > b.l[i]) return 1;
  }
  return 0;
}

inline U256 overflowing_add(const U256& a, const U256& b, bool& of) {
  U256 r;
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a.l[i] + b.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  of = c != 0;
  return r;
}

inline U256 overflowing_sub(const U256& a, const U256& b, bool& of) {
  U256 r;
  uint64_t borrow = 0;
  for (int i = 0; i < 4; i++) {
    unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - borrow;
    r.l[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  of = borrow != 0;
  return r;
}

// 256 x 256 -> 512, schoolbook over u64 limbs
inline void full_mul(const U256& a, const U256& b, uint64_t out[8]) {
  for (int i = 0; i < 8; i++) out[i] = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; j++) {
      unsigned __int128 t = (unsigned __int128)a.l[i] * b.l[j] + out[i + j] + carry;
      out[i + j] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    out[i + 4] = carry;
  }
}

inline U256 shl(const U256& a, uint32_t n) {
  U256 r = U256::zero();
  if (n >= 256) return r;
  uint32_t ws = n / 64, bs = n % 64;
  for (uint32_t i = ws; i < 4; i++) r.l[i] = a.l[i - ws] << bs;
  if (bs > 0)
    for (uint32_t i = ws + 1; i < 4; i++) r.l[i] |= a.l[i - 1 - ws] >> (64 - bs);
  return r;
}

inline U256 shr(const U256& a, uint32_t n) {
  U256 r = U256::zero();
  if (n >= 256) return r;
  uint32_t ws = n / 64, bs = n % 64;
  for (uint32_t i = ws; i < 4; i++) r.l[i - ws] = a.l[i] >> bs;
  if (bs > 0)
    for (uint32_t i = ws + 1; i < 4; i++) r.l[i - ws - 1] |= a.l[i] << (64 - bs);
  return r;
}

inline U256 bit_or(const U256& a, const U256& b) { return U256{{a.l[0] | b.l[0], a.l[1] | b.l[1], a.l[2] | b.l[2], a.l[3] | b.l[3]}}; }
inline U256 bit_and(const U256& a, const U256& b) { return U256{{a.l[0] & b.l[0], a.l[1] & b.l[1], a.l[2] & b.l[2], a.l[3] & b.l[3]}}; }
inline U256 bit_xor(const U256& a, const U256& b) { return U256{{a.l[0] ^ b.l[0], a.l[1] ^ b.l[1], a.l[2] ^ b.l[2], a.l[3] ^ b.l[3]}}; }

// Knuth algorithm D over u64 digits (what the `uint` crate does); b != 0.
inline void div_mod(const U256& a, const U256& b, U256& q, U256& r) {
  q = U256::zero();
  r = U256::zero();
  int n = 4;
  while (n > 0 && b.l[n - 1] == 0) n--;
  int m = 4;
  while (m > 0 && a.l[m - 1] == 0) m--;
  if (m < n) {
    r = a;
    return;
  }
  if (n == 1) {
    unsigned __int128 rem = 0;
    for (int i = m - 1; i >= 0; i--) {
      unsigned __int128 cur = (rem << 64) | a.l[i];
      q.l[i] = (uint64_t)(cur / b.l[0]);
      rem = cur % b.l[0];
    }
    r.l[0] = (uint64_t)rem;
    return;
  }
  const int s = __builtin_clzll(b.l[n - 1]);
  uint64_t vn[4] = {0, 0, 0, 0}, un[5] = {0, 0, 0, 0, 0};
  for (int i = n - 1; i > 0; i--) vn[i] = s ? ((b.l[i] << s) | (b.l[i - 1] >> (64 - s))) : b.l[i];
  vn[0] = b.l[0] << s;
  un[m] = s ? (a.l[m - 1] >> (64 - s)) : 0;
  for (int i = m - 1; i > 0; i--) un[i] = s ? ((a.l[i] << s) | (a.l[i - 1] >> (64 - s))) : a.l[i];
  un[0] = a.l[0] << s;
  const unsigned __int128 B = (unsigned __int128)1 << 64;
  for (int j = m - n; j >= 0; j--) {
    unsigned __int128 num = ((unsigned __int128)un[j + n] << 64) | un[j + n - 1];
    unsigned __int128 qhat = num / vn[n - 1];
    unsigned __int128 rhat = num % vn[n - 1];
    while (qhat >= B || qhat * vn[n - 2] > ((rhat << 64) | un[j + n - 2])) {
      qhat -= 1;
      rhat += vn[n - 1];
      if (rhat >= B) break;
    }
    // multiply and subtract
    __int128 borrow = 0;
    unsigned __int128 carry = 0;
    for (int i = 0; i < n; i++) {
      unsigned __int128 p = qhat * vn[i] + carry;
      carry = p >> 64;
      __int128 t = (__int128)un[i + j] - borrow - (uint64_t)p;
      un[i + j] = (uint64_t)t;
      borrow = t < 0 ? 1 : 0;
    }
    __int128 t = (__int128)un[j + n] - borrow - (uint64_t)carry;
    un[j + n] = (uint64_t)t;
    if (t < 0) {
      qhat -= 1;
      unsigned __int128 c = 0;
      for (int i = 0; i < n; i++) {
        c += (unsigned __int128)un[i + j] + vn[i];
        un[i + j] = (uint64_t)c;
        c >>= 64;
      }
      un[j + n] += (uint64_t)c;
    }
    q.l[j] = (uint64_t)qhat;
  }
  for (int i = 0; i < n - 1; i++) r.l[i] = s ? ((un[i] >> s) | (un[i + 1] << (64 - s))) : un[i];
  r.l[n - 1] = un[n - 1] >> s;
}

// big-endian 32-byte <-> U256 (U256::to_big_endian / from_big_endian)
inline void to_big_endian(const U256& v, uint8_t out[32]) {
  for (int i = 0; i < 4; i++)
    for (int b = 0; b < 8; b++) out[31 - (i * 8 + b)] = (uint8_t)(v.l[i] >> (8 * b));
}
inline U256 from_big_endian(const uint8_t in[32]) {
  U256 v = U256::zero();
  for (int i = 0; i < 4; i++)
    for (int b = 0; b < 8; b++) v.l[i] |= (uint64_t)in[31 - (i * 8 + b)] << (8 * b);
  return v;
}

}  // namespace zko
