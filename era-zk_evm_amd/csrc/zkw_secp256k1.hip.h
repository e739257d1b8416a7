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
ZD u256 ec_mulmod_inl(const u256& a, const u256& b, u32 which) {
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
ZNI u256 ec_mulmod(u256 a, u256 b, u32 which) { return ec_mulmod_inl(a, b, which); }
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

// Points live in the lane's scratch row in global memory (Shared::krow behind the Keccak block: ZKW_EC_SLOTS 256-bit slots,
// dword k of slot i at row[(8 i + k) * L]), not in by-value structs: as `ec_jac` arguments and locals (96 bytes each) they put
// 1184 bytes of frame on the call chain that sizes the cycle kernel's private segment — every launch reserved scratch memory for a
// precompile that a cfg-2 tape never calls.  Performance of this path does not matter (about 6500 modular multiplications).
//   slots 0..2 acc (x, y, z; z == 0: infinity), 3..4 R (affine), 5..7 G + R (Jacobian), 8..9 u1, u2
#define EC_ACC 0u
#define EC_R 3u
#define EC_GR 5u
#define EC_U1 8u
#define EC_U2 9u
ZD u256 ecr_load(const u32* row, u32 L, u32 slot) {
  u256 v;
#pragma unroll
  for (u32 k = 0; k < 8; k++) v.w[k] = row[(u64)(8u * slot + k) * L];
  return v;
}
ZD void ecr_store(u32* row, u32 L, u32 slot, const u256& v) {
#pragma unroll
  for (u32 k = 0; k < 8; k++) row[(u64)(8u * slot + k) * L] = v.w[k];
}
ZD u256 ec_gx() {
  u256 g;
  g.w[0] = 0x16F81798u; g.w[1] = 0x59F2815Bu; g.w[2] = 0x2DCE28D9u; g.w[3] = 0x029BFCDBu; g.w[4] = 0xCE870B07u; g.w[5] = 0x55A06295u; g.w[6] = 0xF9DCBBACu; g.w[7] = 0x79BE667Eu;
  return g;
}
ZD u256 ec_gy() {
  u256 g;
  g.w[0] = 0xFB10D4B8u; g.w[1] = 0x9C47D08Fu; g.w[2] = 0xA6855419u; g.w[3] = 0xFD17B448u; g.w[4] = 0x0E1108A8u; g.w[5] = 0x5DA4FBFCu; g.w[6] = 0x26A3C465u; g.w[7] = 0x483ADA77u;
  return g;
}
#define EC_M(a, b) ec_mulmod(a, b, 0)
#define EC_MI(a, b) ec_mulmod_inl(a, b, 0) /* inside the point operations: inlined — no call, so nothing has to survive one on the stack */
// point at slots [dst, dst + 3) := 2 * point at [src, src + 3)          dbl-2009-l (a = 0)
ZNI void ec_double_row(u32* row, u32 L, u32 dst, u32 src) {
  const u256 P = ec_p();
  const u256 py = ecr_load(row, L, src + 1), pz = ecr_load(row, L, src + 2);
  if (u256_is_zero(pz) || u256_is_zero(py)) {
    ecr_store(row, L, dst, u256_from_u32(1)); ecr_store(row, L, dst + 1, u256_from_u32(1)); ecr_store(row, L, dst + 2, u256_zero());
    return;
  }
  const u256 yz = EC_MI(py, pz);
  ecr_store(row, L, dst + 2, ec_addmod(yz, yz, P));  // (z first: x and y of the source are still in place when dst == src)
  const u256 px = ecr_load(row, L, src);
  const u256 A = EC_MI(px, px), B = EC_MI(py, py), C = EC_MI(B, B);
  const u256 t = ec_addmod(px, B, P);
  u256 D = ec_submod(ec_submod(EC_MI(t, t), A, P), C, P);
  D = ec_addmod(D, D, P);
  const u256 E = ec_addmod(ec_addmod(A, A, P), A, P), F = EC_MI(E, E);
  const u256 rx = ec_submod(F, ec_addmod(D, D, P), P);
  u256 C8 = ec_addmod(C, C, P);
  C8 = ec_addmod(C8, C8, P);
  C8 = ec_addmod(C8, C8, P);
  ecr_store(row, L, dst, rx);
  ecr_store(row, L, dst + 1, ec_submod(EC_MI(E, ec_submod(D, rx, P)), C8, P));
}
// the second operand of an addition: 0 = G (affine), 1 = R (affine, slots EC_R), 2 = G + R (Jacobian, slots EC_GR)
ZD u256 ec_q_coord(const u32* row, u32 L, u32 which, u32 c) {
  if (which == 0u) return c == 0u ? ec_gx() : (c == 1u ? ec_gy() : u256_from_u32(1));
  if (which == 1u) return c == 2u ? u256_from_u32(1) : ecr_load(row, L, EC_R + c);
  return ecr_load(row, L, EC_GR + c);
}
// point at [dst, dst + 3) := point at [p, p + 3) + operand `which`: general Jacobian addition with the equal / opposite cases.
// (One intermediate, S1, waits in slot 10.  Staging every intermediate through the row, or inlining both point operations into the
// caller, made the frames LARGER: what a function of this size keeps on the stack is mostly the callee-saved registers it touches.)
// returns true when the operands are the same point: the caller doubles instead (no call from here: the two frames do not stack)
ZNI bool ec_add_row(u32* row, u32 L, u32 dst, u32 p, u32 which) {
  const u256 P = ec_p();
  const u256 pz = ecr_load(row, L, p + 2), qz = ec_q_coord(row, L, which, 2);
  if (u256_is_zero(pz)) {
    ecr_store(row, L, dst, ec_q_coord(row, L, which, 0)); ecr_store(row, L, dst + 1, ec_q_coord(row, L, which, 1)); ecr_store(row, L, dst + 2, qz);
    return false;
  }
  if (u256_is_zero(qz)) {
    if (dst != p) {
      ecr_store(row, L, dst, ecr_load(row, L, p)); ecr_store(row, L, dst + 1, ecr_load(row, L, p + 1)); ecr_store(row, L, dst + 2, pz);
    }
    return false;
  }
  const u256 Z1Z1 = EC_MI(pz, pz), Z2Z2 = EC_MI(qz, qz);
  const u256 U1 = EC_MI(ecr_load(row, L, p), Z2Z2), U2 = EC_MI(ec_q_coord(row, L, which, 0), Z1Z1);
  const u256 S1 = EC_MI(EC_MI(ecr_load(row, L, p + 1), qz), Z2Z2);
  ecr_store(row, L, 10u, S1);
  const u256 S2 = EC_MI(EC_MI(ec_q_coord(row, L, which, 1), pz), Z1Z1);
  if (u256_eq(U1, U2)) {
    if (u256_eq(S1, S2)) return true;
    ecr_store(row, L, dst, u256_from_u32(1)); ecr_store(row, L, dst + 1, u256_from_u32(1)); ecr_store(row, L, dst + 2, u256_zero());
    return false;
  }
  const u256 H = ec_submod(U2, U1, P), R = ec_submod(S2, ecr_load(row, L, 10u), P);
  ecr_store(row, L, dst + 2, EC_MI(EC_MI(pz, qz), H));  // (x, y of p have been consumed: dst may be p)
  const u256 HH = EC_MI(H, H), HHH = EC_MI(H, HH), V = EC_MI(U1, HH);
  const u256 rx = ec_submod(ec_submod(EC_MI(R, R), HHH, P), ec_addmod(V, V, P), P);
  ecr_store(row, L, dst, rx);
  ecr_store(row, L, dst + 1, ec_submod(EC_MI(R, ec_submod(V, rx, P)), EC_MI(ecr_load(row, L, 10u), HHH), P));
  return false;
}

struct ec_result {
  u256 address_word;  // 12 zero bytes || 20 address bytes as one big-endian word (zero on failure)
  u32 ok;
};

// v_odd: parity of R.y.  Failure (ok = 0) for everything the k256 path reports as Err.  `row`: the lane's scratch row
// (ZKW_EC_SLOTS slots, stride L dwords).
ZD ec_result zkw_ecrecover(u32* row, u32 L, const u256& digest, const u256& r, const u256& s, u32 v_odd) {
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
  ecr_store(row, L, EC_R, r);
  ecr_store(row, L, EC_R + 1, y);
  {
    bool of;
    u256 z = digest;
    if (ec_ge(z, N)) z = u256_sub(z, N, of);  // digest < 2^256 < 2n: one subtraction reduces it
    const u256 nm2 = u256_sub(N, u256_from_u32(2), of);
    const u256 rinv = ec_powmod(r, nm2, 1);
    ecr_store(row, L, EC_U1, ec_mulmod(ec_submod(u256_zero(), z, N), rinv, 1));
    ecr_store(row, L, EC_U2, ec_mulmod(s, rinv, 1));
  }
  // G + R once, then Shamir's double-and-add over (u1, u2)
  ecr_store(row, L, EC_GR, ec_gx()); ecr_store(row, L, EC_GR + 1, ec_gy()); ecr_store(row, L, EC_GR + 2, u256_from_u32(1));
  if (ec_add_row(row, L, EC_GR, EC_GR, 1u)) ec_double_row(row, L, EC_GR, EC_GR);  // (R == G)
  ecr_store(row, L, EC_ACC, u256_from_u32(1)); ecr_store(row, L, EC_ACC + 1, u256_from_u32(1)); ecr_store(row, L, EC_ACC + 2, u256_zero());
#pragma unroll 1
  for (int i = 255; i >= 0; i--) {
    ec_double_row(row, L, EC_ACC, EC_ACC);
    const u32 b1 = (row[(u64)(8u * EC_U1 + (u32)(i >> 5)) * L] >> (i & 31)) & 1u, b2 = (row[(u64)(8u * EC_U2 + (u32)(i >> 5)) * L] >> (i & 31)) & 1u;
    if (b1 | b2)
      if (ec_add_row(row, L, EC_ACC, EC_ACC, b2 ? (b1 ? 2u : 1u) : 0u)) ec_double_row(row, L, EC_ACC, EC_ACC);
  }
  const u256 az = ecr_load(row, L, EC_ACC + 2);
  if (u256_is_zero(az)) return out;
  bool of;
  const u256 pm2 = u256_sub(P, u256_from_u32(2), of);
  const u256 zi = ec_powmod(az, pm2, 0), zi2 = EC_M(zi, zi);
  const u256 ax = EC_M(ecr_load(row, L, EC_ACC), zi2), ay = EC_M(ecr_load(row, L, EC_ACC + 1), EC_M(zi2, zi));
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
