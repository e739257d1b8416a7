// PCIe probe (round 5): what the link between an MI355X and its host sustains, and by which route.
//   D2H / H2D hipMemcpyAsync between device memory and pinned host memory, by transfer size
//   a kernel that stores straight into pinned host memory (coherent / non-coherent mapping), and one that loads from it
// build: hipcc --offload-arch=gfx950 -O3 -o pcie_probe pcie_probe.hip ; run: ./pcie_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                               \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ void copy_kernel(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16, int nt) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const v4u v = src[i];
    if (nt) __builtin_nontemporal_store(v, dst + i);
    else dst[i] = v;
  }
}

static double time_ms(hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  float ms = 0;
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  (void)st;
  return ms;
}

int main() {
  const size_t MAX = 512ull << 20;
  void *dev = nullptr, *dev2 = nullptr, *host_c = nullptr, *host_nc = nullptr;
  CK(hipMalloc(&dev, MAX));
  CK(hipMalloc(&dev2, MAX));
  CK(hipHostMalloc(&host_c, MAX, hipHostMallocDefault));
  CK(hipHostMalloc(&host_nc, MAX, hipHostMallocNonCoherent));
  CK(hipMemset(dev, 1, MAX));
  memset(host_c, 2, MAX);
  memset(host_nc, 3, MAX);
  hipStream_t st, st2;
  CK(hipStreamCreate(&st));
  CK(hipStreamCreate(&st2));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t sizes[] = {256u << 10, 1u << 20, 4u << 20, 16u << 20, 64u << 20, 128u << 20, 256u << 20, 512u << 20};
  for (int dir = 0; dir < 2; dir++) {
    for (int nc = 0; nc < 2; nc++) {
      void* host = nc ? host_nc : host_c;
      for (size_t sz : sizes) {
        const int reps = sz < (16u << 20) ? 50 : 8;
        double best = 1e30;
        for (int t = 0; t < 3; t++) {
          CK(hipEventRecord(e0, st));
          for (int r = 0; r < reps; r++) {
            if (dir == 0) CK(hipMemcpyAsync(host, dev, sz, hipMemcpyDeviceToHost, st));
            else CK(hipMemcpyAsync(dev, host, sz, hipMemcpyHostToDevice, st));
          }
          CK(hipEventRecord(e1, st));
          const double ms = time_ms(st, e0, e1) / reps;
          if (ms < best) best = ms;
        }
        printf("%s memcpyAsync %s pinned  %8.2f MB: %8.3f ms  %6.1f GB/s\n", dir == 0 ? "D2H" : "H2D", nc ? "noncoherent" : "coherent   ", sz / 1048576.0, best, sz / best / 1e6);
      }
    }
  }
  // both directions at once (full duplex)
  {
    const size_t sz = 256u << 20;
    hipEvent_t f0, f1;
    CK(hipEventCreate(&f0));
    CK(hipEventCreate(&f1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st));
    CK(hipEventRecord(f0, st2));
    for (int r = 0; r < 8; r++) {
      CK(hipMemcpyAsync(host_c, dev, sz, hipMemcpyDeviceToHost, st));
      CK(hipMemcpyAsync(dev2, host_nc, sz, hipMemcpyHostToDevice, st2));
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventRecord(f1, st2));
    const double a = time_ms(st, e0, e1) / 8, b = time_ms(st2, f0, f1) / 8;
    printf("duplex 256 MB: D2H %6.1f GB/s  H2D %6.1f GB/s\n", sz / a / 1e6, sz / b / 1e6);
  }
  // kernels that touch host memory directly
  for (int nc = 0; nc < 2; nc++) {
    void* host = nc ? host_nc : host_c;
    void* hd = nullptr;
    CK(hipHostGetDevicePointer(&hd, host, 0));
    for (int nt = 0; nt < 2; nt++)
      for (int blocks : {64, 256, 1024, 4096}) {
        const size_t sz = 256u << 20;
        double best = 1e30;
        for (int t = 0; t < 3; t++) {
          CK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st, (const v4u*)dev, (v4u*)hd, sz / 16, nt);
          CK(hipEventRecord(e1, st));
          const double ms = time_ms(st, e0, e1);
          if (ms < best) best = ms;
        }
        printf("kernel store -> host %s %s %5d blocks: %8.3f ms  %6.1f GB/s\n", nc ? "noncoherent" : "coherent   ", nt ? "nontemporal" : "plain      ", blocks, best, sz / best / 1e6);
      }
    for (int blocks : {256, 1024, 4096}) {
      const size_t sz = 256u << 20;
      double best = 1e30;
      for (int t = 0; t < 3; t++) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st, (const v4u*)hd, (v4u*)dev2, sz / 16, 0);
        CK(hipEventRecord(e1, st));
        const double ms = time_ms(st, e0, e1);
        if (ms < best) best = ms;
      }
      printf("kernel load  <- host %s             %5d blocks: %8.3f ms  %6.1f GB/s\n", nc ? "noncoherent" : "coherent   ", blocks, best, sz / best / 1e6);
    }
  }
  // D2D for reference (what a device-side pack costs)
  {
    const size_t sz = 256u << 20;
    double best = 1e30;
    for (int t = 0; t < 3; t++) {
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, st, (const v4u*)dev, (v4u*)dev2, sz / 16, 1);
      CK(hipEventRecord(e1, st));
      const double ms = time_ms(st, e0, e1);
      if (ms < best) best = ms;
    }
    printf("kernel D2D 256 MB: %8.3f ms  %6.1f GB/s (read + write %6.1f)\n", best, sz / best / 1e6, 2 * sz / best / 1e6);
  }
  return 0;
}
