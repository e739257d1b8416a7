// ORACLE — TEST INFRASTRUCTURE ONLY (see u256.hpp).
//
// secp256k1 public-key recovery for the ecrecover precompile.  The precompile lives in the absent crate
// `zk_evm_abstractions` (branch v1.4.1), which delegates to `k256` (RustCrypto): Signature::from_scalars (r, s in
// [1, n-1]), RecoveryId with y-parity only, VerifyingKey::recover_from_prehash (z = digest reduced mod n,
// Q = r^-1 (s R - z G)), address = keccak256(x || y)[12..].  This file restates that published algorithm in the
// plainest form (Jacobian coordinates, square-and-multiply, Fermat inversions); it is pinned by the two literal
// vectors of the reference's own (stale) test src/testing/tests/precompiles/ecrecover.rs:127-143 and by an
// independent arbitrary-precision Python implementation (tests/secp256k1_ref.py) on random signatures.
#pragma once
#include "hashes.hpp"
#include "u256.hpp"

namespace zko {
namespace secp {

inline U256 hex256(uint64_t a3, uint64_t a2, uint64_t a1, uint64_t a0) { return U256{{a0, a1, a2, a3}}; }
static const U256 P = hex256(0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFEFFFFFC2FULL);
static const U256 N = hex256(0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFEULL, 0xBAAEDCE6AF48A03BULL, 0xBFD25E8CD0364141ULL);
static const U256 CP = hex256(0, 0, 0, 0x1000003D1ULL);                                        // 2^256 - p
static const U256 CN = hex256(0, 0x1ULL, 0x4551231950B75FC4ULL, 0x402DA1732FC9BEBFULL);        // 2^256 - n
static const U256 GX = hex256(0x79BE667EF9DCBBACULL, 0x55A06295CE870B07ULL, 0x029BFCDB2DCE28D9ULL, 0x59F2815B16F81798ULL);
static const U256 GY = hex256(0x483ADA7726A3C465ULL, 0x5DA4FBFC0E1108A8ULL, 0xFD17B448A6855419ULL, 0x9C47D08FFB10D4B8ULL);

struct Modulus {
  const U256& m;  // the modulus, = 2^256 - c
  const U256& c;
};
static const Modulus FP{P, CP}, FN{N, CN};

// (hi : lo) mod m by folding 2^256 = c (mod m) until the high half is empty
inline U256 reduce512(const uint64_t x[8], const Modulus& M) {
  U256 lo{{x[0], x[1], x[2], x[3]}}, hi{{x[4], x[5], x[6], x[7]}};
  while (!hi.is_zero()) {
    uint64_t t[8];
    full_mul(hi, M.c, t);
    U256 tlo{{t[0], t[1], t[2], t[3]}};
    hi = U256{{t[4], t[5], t[6], t[7]}};
    bool of;
    lo = overflowing_add(lo, tlo, of);
    if (of) {
      bool of2;
      hi = overflowing_add(hi, U256::from_u64(1), of2);
    }
  }
  while (cmp(lo, M.m) >= 0) {
    bool of;
    lo = overflowing_sub(lo, M.m, of);
  }
  return lo;
}
inline U256 mulmod(const U256& a, const U256& b, const Modulus& M) {
  uint64_t t[8];
  full_mul(a, b, t);
  return reduce512(t, M);
}
inline U256 addmod(const U256& a, const U256& b, const Modulus& M) {  // a, b < m
  bool of;
  U256 r = overflowing_add(a, b, of);
  if (of || cmp(r, M.m) >= 0) {
    bool of2;
    r = overflowing_sub(r, M.m, of2);
  }
  return r;
}
inline U256 submod(const U256& a, const U256& b, const Modulus& M) {  // a, b < m
  bool of;
  U256 r = overflowing_sub(a, b, of);
  if (of) {
    bool of2;
    r = overflowing_add(r, M.m, of2);
  }
  return r;
}
inline U256 powmod(const U256& a, const U256& e, const Modulus& M) {
  U256 r = U256::from_u64(1);
  for (int i = 255; i >= 0; i--) {
    r = mulmod(r, r, M);
    if ((e.l[i / 64] >> (i % 64)) & 1) r = mulmod(r, a, M);
  }
  return r;
}
inline U256 invmod(const U256& a, const Modulus& M) {  // Fermat: a^(m-2)
  bool of;
  return powmod(a, overflowing_sub(M.m, U256::from_u64(2), of), M);
}

struct Jac {
  U256 x, y, z;  // z == 0: point at infinity
};
inline Jac jac_infinity() { return Jac{U256::from_u64(1), U256::from_u64(1), U256::zero()}; }
inline Jac jac_double(const Jac& p) {  // a = 0: dbl-2009-l
  if (p.z.is_zero() || p.y.is_zero()) return jac_infinity();
  U256 A = mulmod(p.x, p.x, FP), B = mulmod(p.y, p.y, FP), C = mulmod(B, B, FP);
  U256 t = addmod(p.x, B, FP);
  U256 D = submod(submod(mulmod(t, t, FP), A, FP), C, FP);
  D = addmod(D, D, FP);
  U256 E = addmod(addmod(A, A, FP), A, FP), F = mulmod(E, E, FP);
  Jac r;
  r.x = submod(F, addmod(D, D, FP), FP);
  U256 C8 = addmod(C, C, FP);
  C8 = addmod(C8, C8, FP);
  C8 = addmod(C8, C8, FP);
  r.y = submod(mulmod(E, submod(D, r.x, FP), FP), C8, FP);
  U256 yz = mulmod(p.y, p.z, FP);
  r.z = addmod(yz, yz, FP);
  return r;
}
inline Jac jac_add(const Jac& p, const Jac& q) {  // add-2007-bl shape, with the equal / opposite cases
  if (p.z.is_zero()) return q;
  if (q.z.is_zero()) return p;
  U256 Z1Z1 = mulmod(p.z, p.z, FP), Z2Z2 = mulmod(q.z, q.z, FP);
  U256 U1 = mulmod(p.x, Z2Z2, FP), U2 = mulmod(q.x, Z1Z1, FP);
  U256 S1 = mulmod(mulmod(p.y, q.z, FP), Z2Z2, FP), S2 = mulmod(mulmod(q.y, p.z, FP), Z1Z1, FP);
  if (U1 == U2) return S1 == S2 ? jac_double(p) : jac_infinity();
  U256 H = submod(U2, U1, FP), R = submod(S2, S1, FP);
  U256 HH = mulmod(H, H, FP), HHH = mulmod(H, HH, FP), V = mulmod(U1, HH, FP);
  Jac r;
  r.x = submod(submod(mulmod(R, R, FP), HHH, FP), addmod(V, V, FP), FP);
  r.y = submod(mulmod(R, submod(V, r.x, FP), FP), mulmod(S1, HHH, FP), FP);
  r.z = mulmod(mulmod(p.z, q.z, FP), H, FP);
  return r;
}

// returns false on any failure the k256 path reports as Err; address = last 20 bytes of keccak256(x || y)
inline bool ecrecover(const U256& digest, const U256& r, const U256& s, bool v_odd, uint8_t address[20]) {
  if (r.is_zero() || cmp(r, N) >= 0 || s.is_zero() || cmp(s, N) >= 0) return false;  // Signature::from_scalars
  const U256 seven = U256::from_u64(7);
  U256 y2 = addmod(mulmod(mulmod(r, r, FP), r, FP), seven, FP);
  // sqrt: p = 3 (mod 4)  =>  y = y2^((p+1)/4)
  const U256 exp = hex256(0x3FFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFBFFFFF0CULL);
  U256 y = powmod(y2, exp, FP);
  if (mulmod(y, y, FP) != y2) return false;  // x is not on the curve
  if (((y.l[0] & 1) != 0) != v_odd) {
    bool of;
    y = overflowing_sub(P, y, of);
  }
  uint64_t zt[8] = {digest.l[0], digest.l[1], digest.l[2], digest.l[3], 0, 0, 0, 0};
  U256 z = reduce512(zt, FN);
  U256 rinv = invmod(r, FN);
  U256 u1 = mulmod(submod(U256::zero(), z, FN), rinv, FN);  // -z / r
  U256 u2 = mulmod(s, rinv, FN);
  // Shamir: u1 G + u2 R
  Jac G{GX, GY, U256::from_u64(1)}, Rp{r, y, U256::from_u64(1)};
  Jac GR = jac_add(G, Rp);
  Jac acc = jac_infinity();
  for (int i = 255; i >= 0; i--) {
    acc = jac_double(acc);
    int b1 = (u1.l[i / 64] >> (i % 64)) & 1, b2 = (u2.l[i / 64] >> (i % 64)) & 1;
    if (b1 && b2) acc = jac_add(acc, GR);
    else if (b1) acc = jac_add(acc, G);
    else if (b2) acc = jac_add(acc, Rp);
  }
  if (acc.z.is_zero()) return false;
  U256 zi = invmod(acc.z, FP), zi2 = mulmod(zi, zi, FP);
  U256 ax = mulmod(acc.x, zi2, FP), ay = mulmod(acc.y, mulmod(zi2, zi, FP), FP);
  uint8_t pub[64];
  to_big_endian(ax, pub);
  to_big_endian(ay, pub + 32);
  uint8_t h[32];
  keccak256(pub, 64, h);
  std::memcpy(address, h + 12, 20);
  return true;
}

}  // namespace secp
}  // namespace zko
