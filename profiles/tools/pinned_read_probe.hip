// How fast do the host's cores read pinned (hipHostMalloc) memory that a kernel has just written, against ordinary memory?
// (round 5: the replay of a delivered step runs on the host's cores straight out of the pinned ring)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void fill(v4u* dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = v4u{(unsigned)i, 1, 2, 3};
}
static double read_gbps(const uint64_t* p, size_t n8, int threads) {
  std::vector<std::thread> ts;
  std::vector<uint64_t> sums(threads);
  const auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < threads; t++)
    ts.emplace_back([&, t] {
      uint64_t a = 0;
      const size_t lo = n8 * t / threads, hi = n8 * (t + 1) / threads;
      for (size_t i = lo; i < hi; i++) a += p[i] * (2 * (i & 63) + 1);
      sums[t] = a;
    });
  for (auto& t : ts) t.join();
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t a = 0;
  for (auto v : sums) a += v;
  if (a == 42) printf("!");
  return n8 * 8 / s / 1e9;
}
int main() {
  const size_t bytes = 1ull << 30;
  for (unsigned flags : {(unsigned)hipHostMallocDefault, (unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocNumaUser}) {
    void* h = nullptr;
    if (hipHostMalloc(&h, bytes, flags) != hipSuccess) { printf("flags %u: alloc failed\n", flags); continue; }
    void* d = nullptr;
    hipHostGetDevicePointer(&d, h, 0);
    hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, (v4u*)d, bytes / 16);
    hipDeviceSynchronize();
    for (int th : {1, 16, 64, 128}) printf("pinned flags %u, %3d threads: %7.1f GB/s (first read), %7.1f GB/s (second)\n", flags, th, read_gbps((const uint64_t*)h, bytes / 8, th), read_gbps((const uint64_t*)h, bytes / 8, th));
    hipHostFree(h);
  }
  void* m = malloc(bytes);
  memset(m, 1, bytes);
  for (int th : {1, 16, 64, 128}) printf("malloc, %3d threads: %7.1f GB/s\n", th, read_gbps((const uint64_t*)m, bytes / 8, th));
  return 0;
}
