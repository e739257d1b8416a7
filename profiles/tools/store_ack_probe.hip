// Measures what a vector load costs a wave when streaming stores precede it (gfx950: loads and stores share vmcnt
// and are retired in order, so `s_waitcnt vmcnt(0)` for the load also waits for the acknowledgement of the stores).
//   hipcc --offload-arch=gfx950 -O2 store_ack_probe.hip -o store_ack_probe && ./store_ack_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int NSTORES, bool NT>
__global__ void probe(v4u* out, const v4u* in, unsigned long long* clk, int iters) {
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
  v4u* o = out + (size_t)wave * iters * NSTORES * 64 + lane;
  const v4u* src = in + (size_t)wave * 4096 + lane;
  v4u acc = {0, 0, 0, 0};
  unsigned long long t = 0;
  for (int i = 0; i < iters; i++) {
    v4u v = {(unsigned)i, lane, wave, acc.x};
#pragma unroll
    for (int s = 0; s < NSTORES; s++) {
      if (NT) __builtin_nontemporal_store(v, o + (size_t)(i * NSTORES + s) * 64);
      else o[(size_t)(i * NSTORES + s) * 64] = v;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    const v4u r = src[(size_t)((i * 37) & 63) * 64];
    acc += r;                       // dependent use: s_waitcnt vmcnt(0) here
    asm volatile("" : "+v"(acc));
    t += __builtin_readcyclecounter() - t0;
  }
  if (lane == 0) clk[wave] = t;
  if (acc.x == 0xdeadbeef) out[0] = acc;
}
template <int NS, bool NT>
static void run(const char* name, int blocks, v4u* out, v4u* in, unsigned long long* clk, int iters) {
  hipLaunchKernelGGL((probe<NS, NT>), dim3(blocks), dim3(256), 0, 0, out, in, clk, iters);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((probe<NS, NT>), dim3(blocks), dim3(256), 0, 0, out, in, clk, iters);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += (double)x;
  printf("%-28s waves %5d  load wait %8.1f s_memtime ticks per iteration  kernel %.3f ms\n", name, blocks * 4, s / h.size() / iters, ms);
}
int main() {
  const int iters = 256, maxblocks = 256;
  v4u *out, *in; unsigned long long* clk;
  hipMalloc(&out, (size_t)maxblocks * 4 * iters * 12 * 64 * 16);
  hipMalloc(&in, (size_t)maxblocks * 4 * 4096 * 16);
  hipMemset(in, 1, (size_t)maxblocks * 4 * 4096 * 16);
  hipMalloc(&clk, maxblocks * 4 * 8);
  for (int blocks : {1, 256}) {
    run<0, true>("no stores", blocks, out, in, clk, iters);
    run<2, true>("2 nt stores then load", blocks, out, in, clk, iters);
    run<2, false>("2 plain stores then load", blocks, out, in, clk, iters);
    run<11, true>("11 nt stores then load", blocks, out, in, clk, iters);
    run<11, false>("11 plain stores then load", blocks, out, in, clk, iters);
  }
  return 0;
}
