// HBM probe: write-only (nontemporal / plain 16-byte stores), read-only and copy bandwidth of the chip, to know which
// ceiling a write-dominated kernel (the cycle kernel writes ~85 % of its traffic) actually runs against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void fill_nt(v4u* p, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4u v = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(v, p + i);
}
__global__ void fill_plain(v4u* p, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4u v = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void read_only(const v4u* p, size_t n, v4u* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4u acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc ^= p[i];
  if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ void copy(const v4u* s, v4u* d, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(s[i], d + i);
}
// scattered 1-KB pieces: every wave writes 1 KB (64 lanes x 16 B) at a pseudo-random 1-KB-aligned place, like the
// per-wave stream rows of the cycle kernel
__global__ void fill_scatter(v4u* p, size_t n_kb, unsigned rounds) {
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const unsigned waves = (gridDim.x * blockDim.x) >> 6;
  v4u v = {1u, 2u, 3u, lane};
  for (unsigned r = 0; r < rounds; r++) {
    size_t kb = ((size_t)(wave + (size_t)r * waves) * 2654435761u) % n_kb;
    __builtin_nontemporal_store(v, p + kb * 64 + lane);
  }
}
int main() {
  const size_t bytes = (size_t)8 << 30, n = bytes / 16;
  v4u *a, *b;
  hipMalloc((void**)&a, bytes); hipMalloc((void**)&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto&& launch, double moved) {
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < 5; i++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-28s %8.3f ms  %7.2f TB/s\n", name, best, moved / (best * 1e-3) / 1e12);
  };
  for (int blocks : {2048, 8192}) {
    printf("-- %d blocks of 256 threads\n", blocks);
    time("write nt 16B", [&] { fill_nt<<<blocks, 256>>>(a, n); }, (double)bytes);
    time("write plain 16B", [&] { fill_plain<<<blocks, 256>>>(a, n); }, (double)bytes);
    time("read 16B", [&] { read_only<<<blocks, 256>>>(a, n, b); }, (double)bytes);
    time("copy (read + nt write)", [&] { copy<<<blocks, 256>>>(a, b, n); }, 2.0 * bytes);
    time("write nt scattered 1 KB", [&] { fill_scatter<<<blocks, 256>>>(a, bytes / 1024, 256); }, (double)blocks * 4 * 256 * 1024);
  }
  return 0;
}
