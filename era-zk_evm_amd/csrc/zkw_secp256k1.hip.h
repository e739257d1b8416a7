// ecrecover precompile: secp256k1 public-key recovery on 8 x u32 limbs, one signature per lane.
//
// The precompile itself lives in the absent crate `zk_evm_abstractions` (k256: Signature::from_scalars,
// VerifyingKey::recover_from_prehash; address = keccak256(x || y)[12..]); the reference's call site is
// helpers.rs:196-223 and its own (stale) test src/testing/tests/precompiles/ecrecover.rs holds two literal vectors.
// Rare and long (about 6500 modular multiplications): everything heavy is a real (noinline) function so that the
// cycle kernel's hot loop keeps its code size and register allocation; only the lanes that execute an ecrecover
// precompile call enter it.
#pragma once

#define ZNI __device__ __noinline__

ZD u256 ec_p() { u256 r; r.w[0] = 0xFFFFFC2Fu; r.w[1] = 0xFFFFFFFEu; r.w[2] = r.w[3] = r.w[4] = r.w[5] = r.w[6] = r.w[7] = 0xFFFFFFFFu; return r; }
ZD u256 ec_n() {
  u256 r;
  r.w[0] = 0xD0364141u; r.w[1] = 0xBFD25E8Cu; r.w[2] = 0xAF48A03Bu; r.w[3] = 0xBAAEDCE6u; r.w[4] = 0xFFFFFFFEu; r.w[5] = r.w[6] = r.w[7] = 0xFFFFFFFFu;
  return r;
}
ZD u256 ec_cn() {  // 2^256 - n
  u256 r = u256_zero();
  r.w[0] = 0x2FC9BEBFu; r.w[1] = 0x402DA173u; r.w[2] = 0x50B75FC4u; r.w[3] = 0x45512319u; r.w[4] = 1u;
  return r;
}
ZD bool ec_ge(const u256& a, const u256& b) {  // a >= b
  bool of;
  (void)u256_sub(a, b, of);
  return !of;
}

// a * b mod m; which = 0: m = p (field), 1: m = n (group order).  (hi : lo) is folded with 2^256 = c (mod m).
ZNI u256 ec_mulmod(u256 a, u256 b, u32 which) {
  u256 lo, hi;
  u256_mul(a, b, lo, hi);
#pragma unroll 1
  for (int fold = 0; fold < 4; fold++) {
    if (u256_is_zero(hi)) break;
    u256 tlo, thi;
    if (which == 0) {  // c = 2^32 + 977: hi * 977 + (hi << 32)
      u32 carry = 0;
      u32 t[9];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const u64 v = (u64)hi.w[i] * 977u + carry;
        t[i] = (u32)v;
        carry = (u32)(v >> 32);
      }
      t[8] = carry;
      // + (hi << 32): limb i gets hi.w[i - 1]
      u32 c2 = 0;
      u32 s[10];
      s[0] = t[0];
#pragma unroll
      for (int i = 1; i < 9; i++) {
        const u64 v = (u64)t[i] + hi.w[i - 1] + c2;
        s[i] = (u32)v;
        c2 = (u32)(v >> 32);
      }
      s[9] = c2;
#pragma unroll
      for (int i = 0; i < 8; i++) tlo.w[i] = s[i];
      thi = u256_zero();
      thi.w[0] = s[8];
      thi.w[1] = s[9];
    } else {
      u256_mul(hi, ec_cn(), tlo, thi);
    }
    bool of;
    lo = u256_add(lo, tlo, of);
    hi = thi;
    if (of) {
      bool of2;
      hi = u256_add(hi, u256_from_u32(1), of2);
    }
  }
  const u256 m = which == 0 ? ec_p() : ec_n();
  if (ec_ge(lo, m)) {
    bool of;
    lo = u256_sub(lo, m, of);
  }
  return lo;
}
ZD u256 ec_addmod(const u256& a, const u256& b, const u256& m) {  // a, b < m
  bool of, of2;
  u256 r = u256_add(a, b, of);
  if (of || ec_ge(r, m)) r = u256_sub(r, m, of2);
  return r;
}
ZD u256 ec_submod(const u256& a, const u256& b, const u256& m) {  // a, b < m
  bool of, of2;
  u256 r = u256_sub(a, b, of);
  if (of) r = u256_add(r, m, of2);
  return r;
}
ZD u32 ec_bit(const u256& e, int i) {
  u32 limb = e.w[0];
#pragma unroll
  for (int k = 1; k < 8; k++) limb = (i >> 5) == k ? e.w[k] : limb;
  return (limb >> (i & 31)) & 1u;
}
// a^e mod m, square-and-multiply from the top bit
ZNI u256 ec_powmod(u256 a, u256 e, u32 which) {
  u256 r = u256_from_u32(1);
#pragma unroll 1
  for (int i = 255; i >= 0; i--) {
    r = ec_mulmod(r, r, which);
    if (ec_bit(e, i)) r = ec_mulmod(r, a, which);
  }
  return r;
}

struct ec_jac {
  u256 x, y, z;  // z == 0: infinity
};
#define EC_M(a, b) ec_mulmod(a, b, 0)
ZNI ec_jac ec_double(ec_jac p) {  // dbl-2009-l (a = 0)
  const u256 P = ec_p();
  ec_jac r;
  if (u256_is_zero(p.z) || u256_is_zero(p.y)) {
    r.x = u256_from_u32(1); r.y = u256_from_u32(1); r.z = u256_zero();
    return r;
  }
  const u256 A = EC_M(p.x, p.x), B = EC_M(p.y, p.y), C = EC_M(B, B);
  const u256 t = ec_addmod(p.x, B, P);
  u256 D = ec_submod(ec_submod(EC_M(t, t), A, P), C, P);
  D = ec_addmod(D, D, P);
  const u256 E = ec_addmod(ec_addmod(A, A, P), A, P), F = EC_M(E, E);
  r.x = ec_submod(F, ec_addmod(D, D, P), P);
  u256 C8 = ec_addmod(C, C, P);
  C8 = ec_addmod(C8, C8, P);
  C8 = ec_addmod(C8, C8, P);
  r.y = ec_submod(EC_M(E, ec_submod(D, r.x, P)), C8, P);
  const u256 yz = EC_M(p.y, p.z);
  r.z = ec_addmod(yz, yz, P);
  return r;
}
ZNI ec_jac ec_add(ec_jac p, ec_jac q) {  // general Jacobian addition with the equal / opposite cases
  const u256 P = ec_p();
  if (u256_is_zero(p.z)) return q;
  if (u256_is_zero(q.z)) return p;
  const u256 Z1Z1 = EC_M(p.z, p.z), Z2Z2 = EC_M(q.z, q.z);
  const u256 U1 = EC_M(p.x, Z2Z2), U2 = EC_M(q.x, Z1Z1);
  const u256 S1 = EC_M(EC_M(p.y, q.z), Z2Z2), S2 = EC_M(EC_M(q.y, p.z), Z1Z1);
  ec_jac r;
  if (u256_eq(U1, U2)) {
    if (u256_eq(S1, S2)) return ec_double(p);
    r.x = u256_from_u32(1); r.y = u256_from_u32(1); r.z = u256_zero();
    return r;
  }
  const u256 H = ec_submod(U2, U1, P), R = ec_submod(S2, S1, P);
  const u256 HH = EC_M(H, H), HHH = EC_M(H, HH), V = EC_M(U1, HH);
  r.x = ec_submod(ec_submod(EC_M(R, R), HHH, P), ec_addmod(V, V, P), P);
  r.y = ec_submod(EC_M(R, ec_submod(V, r.x, P)), EC_M(S1, HHH), P);
  r.z = EC_M(EC_M(p.z, q.z), H);
  return r;
}

struct ec_result {
  u256 address_word;  // 12 zero bytes || 20 address bytes as one big-endian word (zero on failure)
  u32 ok;
};

// v_odd: parity of R.y.  Failure (ok = 0) for everything the k256 path reports as Err.
ZNI ec_result zkw_ecrecover(u256 digest, u256 r, u256 s, u32 v_odd) {
  ec_result out;
  out.address_word = u256_zero();
  out.ok = 0;
  const u256 P = ec_p(), N = ec_n();
  if (u256_is_zero(r) || ec_ge(r, N) || u256_is_zero(s) || ec_ge(s, N)) return out;
  const u256 y2 = ec_addmod(EC_M(EC_M(r, r), r), u256_from_u32(7), P);
  u256 e;  // (p + 1) / 4
  e.w[0] = 0xBFFFFF0Cu; e.w[1] = 0xFFFFFFFFu; e.w[2] = e.w[3] = e.w[4] = e.w[5] = e.w[6] = 0xFFFFFFFFu; e.w[7] = 0x3FFFFFFFu;
  u256 y = ec_powmod(y2, e, 0);
  if (!u256_eq(EC_M(y, y), y2)) return out;
  if ((y.w[0] & 1u) != (v_odd & 1u)) {
    bool of;
    y = u256_sub(P, y, of);
  }
  bool of;
  u256 z = digest;
  if (ec_ge(z, N)) z = u256_sub(z, N, of);  // digest < 2^256 < 2n: one subtraction reduces it
  u256 nm2 = u256_sub(N, u256_from_u32(2), of);
  const u256 rinv = ec_powmod(r, nm2, 1);
  const u256 u1 = ec_mulmod(ec_submod(u256_zero(), z, N), rinv, 1);
  const u256 u2 = ec_mulmod(s, rinv, 1);
  ec_jac G, Rp;
  G.x.w[0] = 0x16F81798u; G.x.w[1] = 0x59F2815Bu; G.x.w[2] = 0x2DCE28D9u; G.x.w[3] = 0x029BFCDBu; G.x.w[4] = 0xCE870B07u; G.x.w[5] = 0x55A06295u;
  G.x.w[6] = 0xF9DCBBACu; G.x.w[7] = 0x79BE667Eu;
  G.y.w[0] = 0xFB10D4B8u; G.y.w[1] = 0x9C47D08Fu; G.y.w[2] = 0xA6855419u; G.y.w[3] = 0xFD17B448u; G.y.w[4] = 0x0E1108A8u; G.y.w[5] = 0x5DA4FBFCu;
  G.y.w[6] = 0x26A3C465u; G.y.w[7] = 0x483ADA77u;
  G.z = u256_from_u32(1);
  Rp.x = r; Rp.y = y; Rp.z = u256_from_u32(1);
  const ec_jac GR = ec_add(G, Rp);
  ec_jac acc;
  acc.x = u256_from_u32(1); acc.y = u256_from_u32(1); acc.z = u256_zero();
#pragma unroll 1
  for (int i = 255; i >= 0; i--) {
    acc = ec_double(acc);
    const u32 b1 = ec_bit(u1, i), b2 = ec_bit(u2, i);
    if (b1 | b2) {
      ec_jac t = G;
      if (b2) t = b1 ? GR : Rp;
      acc = ec_add(acc, t);
    }
  }
  if (u256_is_zero(acc.z)) return out;
  u256 pm2 = u256_sub(P, u256_from_u32(2), of);
  const u256 zi = ec_powmod(acc.z, pm2, 0), zi2 = EC_M(zi, zi);
  const u256 ax = EC_M(acc.x, zi2), ay = EC_M(acc.y, EC_M(zi2, zi));
  // keccak256(x_be || y_be): one 136-byte block
  u64 st[25];
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    st[j] = ((u64)__builtin_bswap32(ax.w[6 - 2 * j]) << 32) | __builtin_bswap32(ax.w[7 - 2 * j]);
    st[4 + j] = ((u64)__builtin_bswap32(ay.w[6 - 2 * j]) << 32) | __builtin_bswap32(ay.w[7 - 2 * j]);
  }
  st[8] ^= 0x01ull;
  st[16] ^= 0x8000000000000000ull;
  zk_keccak_f1600(st);
  // digest bytes 12..31 -> low 20 bytes of a big-endian word
#pragma unroll
  for (int i = 3; i < 8; i++) {
    const u32 le = (u32)(st[i >> 1] >> (32 * (i & 1)));
    out.address_word.w[7 - i] = __builtin_bswap32(le);
  }
  out.ok = 1;
  return out;
}
#undef EC_M
