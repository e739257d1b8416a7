// What a vector store costs the wave that issues it on gfx950 (issue only: the stores are not waited for inside the timed
// block), by width and cache policy, with 1 / 2 waves per SIMD and with every CU busy or one wave alone on the chip.
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/store_issue_probe.hip -o /tmp/store_issue_probe && /tmp/store_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

// MODE 0: dwordx4 nt, 1: dwordx4 plain, 2: dwordx2 nt, 3: dword nt, 4: dwordx4 nt with VALU work between the stores (8 v_add each)
template <int MODE, int NST>
__global__ void probe(char* out, u64* clk, int iters, size_t wave_bytes) {
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
  char* base = out + (size_t)wave * wave_bytes;
  u32 a = lane, b = wave;
  u64 t = 0;
  for (int i = 0; i < iters; i++) {
    char* p = base + (size_t)(i & 63) * NST * 1024;
    const u64 t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int s = 0; s < NST; s++) {
      if (MODE == 0 || MODE == 4) { v4u v = {a, b, (u32)i, (u32)s}; __builtin_nontemporal_store(v, (v4u*)(p + s * 1024) + lane); }
      if (MODE == 1) { v4u v = {a, b, (u32)i, (u32)s}; *((v4u*)(p + s * 1024) + lane) = v; }
      if (MODE == 2) { v2u v = {a, b}; __builtin_nontemporal_store(v, (v2u*)(p + s * 1024) + lane); }
      if (MODE == 3) { __builtin_nontemporal_store(a, (u32*)(p + s * 1024) + lane); }
      if (MODE == 4) { asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
    }
    const u64 t1 = __builtin_readcyclecounter();
    t += t1 - t0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // spacing: the kernel's stores are ~1000 clocks apart
    for (int w = 0; w < 8; w++) asm volatile("s_sleep 8");
  }
  if (lane == 0) clk[wave] = t;
  if (a == 0xdeadbeef) out[0] = (char)b;
}

template <int MODE, int NST>
static void run(const char* name, int threads, int blocks, char* d_out, u64* d_clk, size_t wave_bytes) {
  const int iters = 200;
  hipLaunchKernelGGL((probe<MODE, NST>), dim3(blocks), dim3(threads), 0, 0, d_out, d_clk, 8, wave_bytes);
  hipLaunchKernelGGL((probe<MODE, NST>), dim3(blocks), dim3(threads), 0, 0, d_out, d_clk, iters, wave_bytes);
  CK(hipDeviceSynchronize());
  const int nw = blocks * threads / 64;
  std::vector<u64> h(nw);
  CK(hipMemcpy(h.data(), d_clk, nw * sizeof(u64), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double per = (double)NST * iters;
  printf("%-34s stores per burst %2d  waves %5d: clocks per store instruction min %.1f median %.1f max %.1f\n", name, NST, nw, h[0] / per, h[nw / 2] / per, h[nw - 1] / per);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t wave_bytes = 64 * 16 * 1024;  // 64 slots x 16 stores x 1 KB
  const int max_waves = 2048;
  char* d_out; u64* d_clk;
  CK(hipMalloc(&d_out, wave_bytes * max_waves)); CK(hipMalloc(&d_clk, max_waves * 8));
#define ALL(T, B, TAG) \
  run<0, 1>("dwordx4 nt " TAG, T, B, d_out, d_clk, wave_bytes); run<0, 3>("dwordx4 nt " TAG, T, B, d_out, d_clk, wave_bytes); run<0, 12>("dwordx4 nt " TAG, T, B, d_out, d_clk, wave_bytes); \
  run<1, 3>("dwordx4 plain " TAG, T, B, d_out, d_clk, wave_bytes); run<1, 12>("dwordx4 plain " TAG, T, B, d_out, d_clk, wave_bytes); \
  run<2, 12>("dwordx2 nt " TAG, T, B, d_out, d_clk, wave_bytes); run<3, 12>("dword nt " TAG, T, B, d_out, d_clk, wave_bytes); \
  run<4, 3>("dwordx4 nt + 8 VALU " TAG, T, B, d_out, d_clk, wave_bytes); run<4, 12>("dwordx4 nt + 8 VALU " TAG, T, B, d_out, d_clk, wave_bytes);
  ALL(64, 1, "(one wave on the chip)")
  ALL(256, 256, "(1 wave per SIMD)")
  ALL(320, 256, "(5 waves per CU)")
  ALL(512, 256, "(2 waves per SIMD)")
  return 0;
}
