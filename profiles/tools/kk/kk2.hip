#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32; typedef unsigned long long u64;
#define ZD static __device__ __forceinline__
__device__ const u64 ZKW_KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
ZD u64 zk_rotl64(u64 x, int n) {
  u32 lo = (u32)x, hi = (u32)(x >> 32);
  if (n >= 32) { const u32 t = lo; lo = hi; hi = t; n -= 32; }
  if (n == 0) return ((u64)hi << 32) | lo;
  return ((u64)__builtin_amdgcn_alignbit(hi, lo, (u32)(32 - n)) << 32) | (u64)__builtin_amdgcn_alignbit(lo, hi, (u32)(32 - n));
}
#include "body.h"
#include "body2.h"
template <int V>
__global__ void __launch_bounds__(256, 2) k(u64* p, int n) {
  u64 a[25];
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < 25; i++) a[i] = p[t * 25 + i];
  if (V == 1) { a[1] = ~a[1]; a[2] = ~a[2]; a[8] = ~a[8]; a[12] = ~a[12]; a[17] = ~a[17]; a[20] = ~a[20]; }
  for (int i = 0; i < n; i++) { if (V == 0) zk_keccak_f1600(a); else zk_keccak_f1600_lc(a); }
  if (V == 1) { a[1] = ~a[1]; a[2] = ~a[2]; a[8] = ~a[8]; a[12] = ~a[12]; a[17] = ~a[17]; a[20] = ~a[20]; }
  for (int i = 0; i < 25; i++) p[t * 25 + i] = a[i];
}
int main() {
  const int blocks = 512, n = 200;
  size_t N = (size_t)blocks * 256 * 25;
  std::vector<u64> h(N), r0(N), r1(N);
  for (size_t i = 0; i < N; i++) h[i] = i * 0x9e3779b97f4a7c15ULL + 12345;
  u64* d; hipMalloc(&d, N * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 2; v++) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemcpy(d, h.data(), N * 8, hipMemcpyHostToDevice);
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, n); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("variant %d: %.3f ms  %.2f G keccak-f/s\n", v, ms, (double)blocks * 256 * n / ms / 1e6);
    }
    hipMemcpy(v == 0 ? r0.data() : r1.data(), d, N * 8, hipMemcpyDeviceToHost);
  }
  size_t bad = 0; for (size_t i = 0; i < N; i++) bad += r0[i] != r1[i];
  printf("mismatches %zu\n", bad);
  return 0;
}
