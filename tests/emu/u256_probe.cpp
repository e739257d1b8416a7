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
}
