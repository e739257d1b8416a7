// TEST INFRASTRUCTURE ONLY — exposes the 256-bit helpers of the product header (csrc/zkw_u256.hip.h, compiled by g++
// against the HIP stand-in) so that tests/test_u256_product.py can compare them with Python integers over their whole
// parameter ranges (every shift amount, every byte offset).
#include <hip/hip_runtime.h>
#include "zkw_u256.hip.h"
dim3 threadIdx, blockIdx, blockDim, gridDim;
extern "C" {
void t_shl(const u32* a, u32 n, u32* out) { u256 x; for (int i = 0; i < 8; i++) x.w[i] = a[i]; const u256 r = u256_shl(x, n); for (int i = 0; i < 8; i++) out[i] = r.w[i]; }
void t_shr(const u32* a, u32 n, u32* out) { u256 x; for (int i = 0; i < 8; i++) x.w[i] = a[i]; const u256 r = u256_shr(x, n); for (int i = 0; i < 8; i++) out[i] = r.w[i]; }
void t_low_mask(u32 nbits, u32* out) { const u256 r = u256_low_mask(nbits); for (int i = 0; i < 8; i++) out[i] = r.w[i]; }
void t_window(const u32* hi, const u32* lo, u32 unal, u32* out) {
  u256 h, l;
  for (int i = 0; i < 8; i++) { h.w[i] = hi[i]; l.w[i] = lo[i]; }
  const u256 r = u256_byte_window(h, l, unal);
  for (int i = 0; i < 8; i++) out[i] = r.w[i];
}
static u256 ld(const u32* a) { u256 x; for (int i = 0; i < 8; i++) x.w[i] = a[i]; return x; }
static void st(const u256& r, u32* out) { for (int i = 0; i < 8; i++) out[i] = r.w[i]; }
u32 t_add(const u32* a, const u32* b, u32* out) { bool of; st(u256_add(ld(a), ld(b), of), out); return of; }
u32 t_sub(const u32* a, const u32* b, u32* out) { bool of; st(u256_sub(ld(a), ld(b), of), out); return of; }
void t_mul(const u32* a, const u32* b, u32* lo, u32* hi) { u256 l, h; u256_mul(ld(a), ld(b), l, h); st(l, lo); st(h, hi); }
void t_divmod(const u32* a, const u32* b, u32* q, u32* r) { u256 qq, rr; u256_divmod(ld(a), ld(b), qq, rr); st(qq, q); st(rr, r); }
void t_select_bits(const u32* m, const u32* a, const u32* b, u32* out) {
  u256 mm, aa, bb;
  for (int i = 0; i < 8; i++) { mm.w[i] = m[i]; aa.w[i] = a[i]; bb.w[i] = b[i]; }
  const u256 r = u256_select_bits(mm, aa, bb);
  for (int i = 0; i < 8; i++) out[i] = r.w[i];
}
// the uniform-offset forms (offset the same for every lane: dword part as a template parameter, byte part a scalar)
void t_window_at(const u32* hi, const u32* lo, u32 unal, u32* out) {
  const u256 h = ld(hi), l = ld(lo);
  const u32 b8 = (unal & 3u) * 8u;
  u256 r;
  switch (unal >> 2) {
    case 0: r = u256_byte_window_at<0>(h, l, b8); break; case 1: r = u256_byte_window_at<1>(h, l, b8); break; case 2: r = u256_byte_window_at<2>(h, l, b8); break; case 3: r = u256_byte_window_at<3>(h, l, b8); break;
    case 4: r = u256_byte_window_at<4>(h, l, b8); break; case 5: r = u256_byte_window_at<5>(h, l, b8); break; case 6: r = u256_byte_window_at<6>(h, l, b8); break; default: r = u256_byte_window_at<7>(h, l, b8); break;
  }
  st(r, out);
}
void t_merge_at(const u32* w0, const u32* w1, const u32* v, u32 unal, u32* n0, u32* n1) {
  const u256 a = ld(w0), b = ld(w1), x = ld(v);
  const u32 b8 = (unal & 3u) * 8u;
  u256 r0, r1;
  switch (unal >> 2) {
    case 0: u256_merge_at<0>(a, b, x, b8, r0, r1); break; case 1: u256_merge_at<1>(a, b, x, b8, r0, r1); break; case 2: u256_merge_at<2>(a, b, x, b8, r0, r1); break;
    case 3: u256_merge_at<3>(a, b, x, b8, r0, r1); break; case 4: u256_merge_at<4>(a, b, x, b8, r0, r1); break; case 5: u256_merge_at<5>(a, b, x, b8, r0, r1); break;
    case 6: u256_merge_at<6>(a, b, x, b8, r0, r1); break; default: u256_merge_at<7>(a, b, x, b8, r0, r1); break;
  }
  st(r0, n0); st(r1, n1);
}
}
