// Throughput of the ZKW-GL-sponge permutation (csrc/zkw_goldilocks.hip.h) on the whole chip — every lane chains
// `iters` permutations of its own state — next to the issue rate of v_mad_u64_u32 (the 32 x 32 -> 64 multiply-add the
// field multiplication is made of) and of a plain 32-bit add, which bound it.
//   hipcc --offload-arch=gfx950 -O3 -I era-zk_evm_amd/csrc profiles/tools/perm_probe.hip -o /tmp/perm_probe && /tmp/perm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "zkw_goldilocks.hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) perm_kernel(const u64* rc, u64* out, int iters) {
  u64 s[12];
  const u64 t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = t * 12 + i;
  for (int k = 0; k < iters; k++) gl_permute(rc, s);
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) acc ^= s[i];
  out[t] = acc;
}
// 8 independent chains of v_mad_u64_u32 per lane
__global__ void __launch_bounds__(256) mad_kernel(u64* out, int iters) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  u64 a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = t + i;
  u32 m = t | 1u;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (u64)(u32)a[i] * m + a[i];
  }
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) acc ^= a[i];
  out[t] = acc;
}
__global__ void __launch_bounds__(256) add_kernel(u64* out, int iters) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  u32 a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = t + i;
  u32 m = t | 1u;
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (a[i] ^ m) + (a[i] >> 3);  // 3 full-rate ops (v_xor, v_lshrrev, v_add) — or fused
  }
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) acc ^= a[i];
  out[t] = acc;
}
template <class F>
static float timed(F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 64;
  std::vector<u64> h(118);
  u64 x = 0x123456789abcdefULL;
  for (auto& v : h) { x = x * 6364136223846793005ULL + 1442695040888963407ULL; v = x % GL_P; }
  u64 *rc, *out;
  CK(hipMalloc(&rc, 118 * 8)); CK(hipMemcpy(rc, h.data(), 118 * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)16384 * 256 * 8));
  for (int wgs : {256, 512, 1024, 2048, 4096, 8192}) {  // 256 workgroups of 4 waves = one wave per SIMD: the latency of a lone chain
    const float ms = timed([&] { perm_kernel<<<wgs, 256>>>(rc, out, iters); });
    printf("perm: wgs %d x 256 lanes x %d perms: %.3f ms  %.3f G perms/s\n", wgs, iters, ms, (double)wgs * 256 * iters / ms / 1e6);
  }
  {
    const int n = 4096;
    const float ms = timed([&] { mad_kernel<<<8192, 256>>>(out, n); });
    printf("v_mad_u64_u32: %.3f ms  %.2f T mad/s (lane ops)\n", ms, 8192.0 * 256 * n * 8 / ms / 1e9);
    const float ms2 = timed([&] { add_kernel<<<8192, 256>>>(out, n); });
    printf("32-bit alu (xor/shift/add, 3 ops or fewer per step): %.3f ms  %.2f T steps/s\n", ms2, 8192.0 * 256 * n * 8 / ms2 / 1e9);
  }
  return 0;
}
