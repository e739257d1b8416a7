// What one wave pays per instruction on gfx950, by instruction class and by how many waves share its SIMD — the inputs
// for reading the cycle kernel's phase profile (a VM cycle of a wave is ~900 instructions and ~10k clocks).
// Every kernel times a body with s_memtime inside ONE launch (clocks per body instruction, minimum over the waves),
// with 1 / 2 waves per SIMD (blocks of 256 / 512 threads, one block per CU).
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

#define PROBE(NAME, N_PER_ITER, BODY)                                                        \
  __global__ void NAME(u64* out, int iters, u32* sink, const u32* src) {                      \
    u32 a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = 7u, d = 9u;                            \
    extern __shared__ u32 lds[];                                                               \
    lds[threadIdx.x] = threadIdx.x * 4u;                                                       \
    __syncthreads();                                                                           \
    u32 la = (threadIdx.x & 63u) * 4u;                                                         \
    const u64 t0 = __builtin_readcyclecounter();                                               \
    for (int k = 0; k < iters; k++) { BODY }                                                   \
    const u64 t1 = __builtin_readcyclecounter();                                               \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + la;                           \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
  }                                                                                            \
  static const int NAME##_n = N_PER_ITER;

// dependent VALU chain
PROBE(k_valu_dep, 256, asm volatile(REP256("v_add_u32 %0, %0, %1\n\t") : "+v"(a) : "v"(b));)
// independent VALU (4 chains)
PROBE(k_valu_ind, 256, asm volatile(REP64("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t") : "+v"(a), "+v"(c), "+v"(d), "+v"(la) : "v"(b));)
// dependent SALU chain
PROBE(k_salu_dep, 256, { u32 s = (u32)k; asm volatile(REP256("s_add_u32 %0, %0, 3\n\t") : "+s"(s) : : "scc"); a += s; })
// VALU -> SGPR (readlane) -> SALU -> VALU round trips (the shape of spilled-scalar reloads and scalar decode)
PROBE(k_readlane, 192, { u32 s; asm volatile(REP64("v_readlane_b32 %1, %0, 3\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %1, %0\n\t") : "+v"(a), "=&s"(s) : : "scc"); })
// v_cmp -> vcc -> v_cndmask (vector-written scalar read by vector: wait states)
PROBE(k_cmp_cnd, 128, asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(a) : "v"(b) : "vcc");)
// carry chain
PROBE(k_carry, 128, asm volatile(REP64("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc\n\t") : "+v"(a), "+v"(c) : "v"(b) : "vcc");)
// taken forward branches: 64 x (s_branch over one instruction + 3 VALU)
PROBE(k_branch, 256, asm volatile(REP64("s_branch 1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(a) : "v"(b));)
// untaken conditional branches (scc known 0)
PROBE(k_cbranch_nt, 256, asm volatile("s_cmp_eq_u32 0, 1\n\t" REP64("s_cbranch_scc1 1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(a) : "v"(b) : "scc");)
// exec-mask regions: s_and_saveexec / s_or exec around 2 VALU (the divergent-if pattern), never skipping
PROBE(k_saveexec, 256, { u64 sv; asm volatile(REP64("s_and_saveexec_b64 %2, -1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\ts_or_b64 exec, exec, %2\n\t") : "+v"(a), "+v"(b), "=&s"(sv) : : "scc"); })
// LDS dependent round trips (address from the previous read)
PROBE(k_lds_dep, 64, asm volatile(REP64("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(la));)
// LDS b128 round trip
PROBE(k_lds128_dep, 64, { asm volatile(REP64("ds_read_b128 v[100:103], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xfc, v100\n\t") : "+v"(la) : : "v100", "v101", "v102", "v103"); })
// s_set_gpr_idx_on + 8 movs + off (the register-file access)
PROBE(k_gpridx, 640, { u32 off = 8u * (1u + ((u32)k & 7u)); asm volatile(REP64("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 v104, v128\n\tv_mov_b32 v105, v129\n\tv_mov_b32 v106, v130\n\tv_mov_b32 v107, v131\n\tv_mov_b32 v108, v132\n\tv_mov_b32 v109, v133\n\tv_mov_b32 v110, v134\n\tv_mov_b32 v111, v135\n\ts_set_gpr_idx_off\n\t") : "+v"(a) : "s"(off) : "m0", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v255"); })
// the same through v_movrels-free alternative: 8 x v_readlane-free plain movs (baseline for the pair above)
PROBE(k_mov8, 512, asm volatile(REP64("v_mov_b32 v104, v96\n\tv_mov_b32 v105, v97\n\tv_mov_b32 v106, v98\n\tv_mov_b32 v107, v99\n\tv_mov_b32 v108, v100\n\tv_mov_b32 v109, v101\n\tv_mov_b32 v110, v102\n\tv_mov_b32 v111, v103\n\t") : "+v"(a) : : "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103");)
// global load round trip, L2-resident (dependent address)
PROBE(k_gload_dep, 16, { const u32* p = src; for (int q = 0; q < 16; q++) { u32 v = __builtin_nontemporal_load(p + (a & 1023u)); a = v + (u32)q; } })
// s_load round trip (scalar cache hit)
PROBE(k_sload_dep, 16, { u32 s = (u32)k & 15u; for (int q = 0; q < 16; q++) { u32 v; asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(src), "s"(s * 4u)); s = v & 15u; } a += s; })

// A long straight-line body (instruction-cache footprint): SIZE_K x 1024 dependent v_add (8 bytes each as VOP3? no: 4 B
// VOP2) — use the 8-byte encoding v_add3_u32 to reach a footprint quickly.  16K x 8 B = 128 KB per pass.
#define BIGBODY(NAME, REPS)                                                                    \
  __global__ void NAME(u64* out, int iters, u32* sink, const u32* src) {                       \
    u32 a = threadIdx.x, b = threadIdx.x * 3u + 1u;                                              \
    const u64 t0 = __builtin_readcyclecounter();                                                \
    for (int k = 0; k < iters; k++) { asm volatile(REPS("v_add3_u32 %0, %0, %1, 1\n\t") : "+v"(a) : "v"(b)); } \
    const u64 t1 = __builtin_readcyclecounter();                                                \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;                                             \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
  }
#define REP1K(x) REP256(x) REP256(x) REP256(x) REP256(x)
#define REP4K(x) REP1K(x) REP1K(x) REP1K(x) REP1K(x)
#define REP16K(x) REP4K(x) REP4K(x) REP4K(x) REP4K(x)
BIGBODY(k_code_8k, REP1K)     // 8 KB of code
static const int k_code_8k_n = 1024;
BIGBODY(k_code_32k, REP4K)    // 32 KB
static const int k_code_32k_n = 4096;
BIGBODY(k_code_128k, REP16K)  // 128 KB
static const int k_code_128k_n = 16384;

typedef void (*kern_t)(u64*, int, u32*, const u32*);
static void run(const char* name, kern_t f, int n_per_iter, int iters, int threads, int blocks, u64* d_out, u32* d_sink, const u32* d_src) {
  hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 4096, 0, d_out, 2, d_sink, d_src);  // warm the instruction cache path
  hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 4096, 0, d_out, iters, d_sink, d_src);
  CK(hipDeviceSynchronize());
  const int nw = blocks * threads / 64;
  std::vector<u64> h(nw);
  CK(hipMemcpy(h.data(), d_out, nw * sizeof(u64), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double per = (double)n_per_iter * iters;
  printf("%-14s waves/SIMD %d  blocks %4d: clocks per instruction min %.2f median %.2f max %.2f\n", name, threads / 256, blocks, h[0] / per, h[nw / 2] / per, h[nw - 1] / per);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const char* only = argc > 1 ? argv[1] : nullptr;
  u64* d_out; u32* d_sink; u32* d_src;
  CK(hipMalloc(&d_out, 1 << 20)); CK(hipMalloc(&d_sink, 64 << 20)); CK(hipMalloc(&d_src, 1 << 16));
  std::vector<u32> hs(16384);
  for (int i = 0; i < 16384; i++) hs[i] = (u32)(i * 2654435761u) & 1023u;
  CK(hipMemcpy(d_src, hs.data(), 65536, hipMemcpyHostToDevice));
#define RUN(K, IT) \
  if (!only || !strcmp(only, #K)) { \
  run(#K, K, K##_n, IT, 256, 256, d_out, d_sink, d_src); \
  run(#K, K, K##_n, IT, 512, 256, d_out, d_sink, d_src); \
  if (strcmp(#K, "k_gpridx")) run(#K, K, K##_n, IT, 1024, 256, d_out, d_sink, d_src); }
  RUN(k_valu_dep, 200) RUN(k_valu_ind, 200) RUN(k_salu_dep, 200) RUN(k_readlane, 200) RUN(k_cmp_cnd, 200) RUN(k_carry, 200)
  RUN(k_branch, 200) RUN(k_cbranch_nt, 200) RUN(k_saveexec, 200) RUN(k_lds_dep, 200) RUN(k_lds128_dep, 200) RUN(k_gpridx, 100) RUN(k_mov8, 100)
  RUN(k_gload_dep, 200) RUN(k_sload_dep, 200)
  RUN(k_code_8k, 100) RUN(k_code_32k, 50) RUN(k_code_128k, 20)
  return 0;
}
