// How many workgroups of a given shape (threads, dynamic LDS, registers) are resident at once on this GPU?
// Every workgroup spins for a fixed number of clock ticks; the launch takes (rounds x spin), and the
// HW_ID register tells which XCC / SE / CU each workgroup ran on.   hipcc --offload-arch=gfx950 -O2 occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

template <int REGS, int BIG>
__global__ void __launch_bounds__(256) probe(unsigned long long spin, unsigned* ids, float* sink) {
  extern __shared__ float lds[];
  float acc[REGS];
#pragma unroll
  for (int i = 0; i < REGS; i++) acc[i] = threadIdx.x * 0.5f + i;
  // BIG: force the register allocation of the cycle kernel (1 = 256 VGPR, 2 = 256 VGPR + 174 AGPR, 3 = 256 + 256)
  if (BIG >= 1) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (BIG == 2) asm volatile("v_accvgpr_write_b32 a173, 0" ::: "a173");
  if (BIG == 3) asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
  const unsigned long long t0 = wall_clock64();
  lds[threadIdx.x] = acc[0];
  while (wall_clock64() - t0 < spin) {
#pragma unroll
    for (int i = 0; i < REGS; i++) acc[i] = acc[i] * 1.0001f + acc[(i + 1) % REGS];
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < REGS; i++) s += acc[i];
  if (s == 123.456f) sink[0] = s + lds[0];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ids[blockIdx.x * 2] = hw;
    ids[blockIdx.x * 2 + 1] = xcc;
  }
}

template <int REGS, int BIG>
void run(const char* name, int lds_bytes, int threads) {
  unsigned* ids;
  float* sink;
  hipMalloc(&ids, 8192 * 8);
  hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)probe<REGS, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const unsigned long long spin = 100000;  // 1 ms at the 100 MHz wall clock
  for (int wgs : {256, 320, 512, 640, 768, 1024}) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<REGS, BIG>), dim3(wgs), dim3(threads), lds_bytes, 0, spin, ids, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<REGS, BIG>), dim3(wgs), dim3(threads), lds_bytes, 0, spin, ids, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(wgs * 2);
    hipMemcpy(h.data(), ids, wgs * 8, hipMemcpyDeviceToHost);
    std::set<unsigned long long> cus;
    std::set<unsigned> xccs;
    for (int i = 0; i < wgs; i++) {
      const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
      // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
      cus.insert(((unsigned long long)xcc << 32) | (hw & 0xff00u));
      xccs.insert(xcc);
    }
    printf("%s lds=%d threads=%d wgs=%d: %.2f ms  (%.1f rounds)  distinct CUs %zu on %zu XCCs\n", name, lds_bytes, threads, wgs, ms, ms / 1.0, cus.size(), xccs.size());
  }
  hipFree(ids);
  hipFree(sink);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs, LDS/block %zu, regs/block %d, maxThreadsPerMP %d\n", p.name, p.multiProcessorCount, p.sharedMemPerBlock, p.regsPerBlock, p.maxThreadsPerMultiProcessor);
  // round 2: the cycle kernel is 256 threads, 256 unified registers (v0..v127 compiled + v128..v255 register file), 73,856 B
  // of dynamic LDS: two workgroups per CU = two waves per SIMD -> 512 workgroups (2048 waves) resident
  run<8, 1>("r02 cycle kernel shape (256v, 73856 B LDS)", 73856, 256);
  run<8, 1>("256v, 41 KB LDS", 41984, 256);
  // round 1 shapes, for reference
  run<8, 2>("r01: 256v+174a, 139328 B LDS", 139328, 256);
  run<8, 3>("256v+256a", 1024, 256);
  return 0;
}
