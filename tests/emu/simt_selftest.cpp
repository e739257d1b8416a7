// TEST INFRASTRUCTURE ONLY — the SIMT engine of tests/emu (emu_simt.cpp) tested on its own: small kernels whose results are known
// in closed form exercise what the emulation of the product kernels relies on — which lanes take part in a cross-lane operation
// under ZKW_DIV_IF / else / ZKW_DIV_SCOPE, lanes that leave a scope early, loops with per-lane trip counts, nested scopes, the
// wave's cursor registers, shuffles, a workgroup barrier over `__shared__` memory — and that a cross-lane operation inside a
// divergent region WITHOUT an annotation is reported (mode "unannotated": the process must abort).
// Built and run by tests/test_emu_simt_engine.py:  g++ -DZKW_EMU_WAVE=64 -I tests/emu simt_selftest.cpp emu_simt.cpp emu_glue.cpp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

static int failures = 0;
#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      failures++;                                                       \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);    \
    }                                                                   \
  } while (0)

static unsigned long long out64[1024];
static uint32_t out32[1024];

// (a) if / else: each side's ballot sees exactly its own lanes; behind the scope everyone is back
static void k_if_else(int) {
  const unsigned t = threadIdx.x, lane = t & 63u;
  unsigned long long m = 0;
  ZKW_DIV_IF(lane % 3 == 0) { m = __ballot(1); } else { m = __ballot(lane & 1); }
  out64[t] = m;
  out64[512 + t] = __ballot(1);
}
// (b) a loop whose trip count differs per lane, a ballot in every iteration; lanes that are through wait behind the scope
static void k_ragged_loop(int) {
  const unsigned t = threadIdx.x, lane = t & 63u, n = lane % 5;
  unsigned seen = 0;
  {
    ZKW_DIV_SCOPE;
    for (unsigned i = 0; i < n; i++) seen += (unsigned)__popcll(__ballot(1));  // lanes with n > i are in iteration i
  }
  out32[t] = seen;
  out64[t] = __ballot(1);
}
// (c) early exit from a nested scope + stream-cursor allocation by the lanes that stay
static void k_early_exit(int) {
  const unsigned t = threadIdx.x, lane = t & 63u;
  if (lane == 0) zkw_emu_wave_sregs()[2] = 100;
  __builtin_amdgcn_wave_barrier();
  out32[t] = 0xffffffffu;
  ZKW_DIV_IF(lane >= 8) {
    ZKW_DIV_SCOPE;
    if (lane & 1) return;  // odd lanes leave the kernel from inside two scopes
    const unsigned long long m = __ballot(1);
    const uint32_t base = ZKW_EMU_FETCH_ADD(2, (uint32_t)__popcll(m));
    out32[t] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  out64[t] = __ballot(1);  // the even lanes and lanes 0..7
}
// (d) butterfly maximum + readlane / readfirstlane
static void k_shuffles(int) {
  const unsigned t = threadIdx.x, lane = t & 63u;
  unsigned v = (lane * 37u) % 64u;
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off);
    v = o > v ? o : v;
  }
  out32[t] = v;
  out32[256 + t] = (uint32_t)__builtin_amdgcn_readlane((int)(lane * 3u), 21);
  ZKW_DIV_IF(lane >= 40) out32[512 + t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lane);
}
// (e) workgroup barrier over shared memory: 4 waves, every thread reads what a thread of another wave wrote
static void k_barrier(int) {
  __shared__ uint32_t buf[256];
  const unsigned t = threadIdx.x;
  buf[t] = t * t;
  __syncthreads();
  out32[t] = buf[255 - t];
  __syncthreads();
  buf[t] = 7;
  __syncthreads();
  out32[256 + t] = buf[(t + 64) & 255];
}
// (f) a ballot inside a divergent `if` that is NOT annotated: the lanes that skip it arrive at another cross-lane operation
static void k_unannotated(int) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned long long m = 0;
  if (lane < 10) m = __ballot(1);
  out64[threadIdx.x] = m + __ballot(1);
}

int main(int argc, char** argv) {
  const bool unannotated = argc > 1 && !strcmp(argv[1], "unannotated");
  if (unannotated) {
    hipLaunchKernelGGL(k_unannotated, dim3(1), dim3(64), 0, nullptr, 0);
    printf("NOT DETECTED\n");
    return 0;
  }
  hipLaunchKernelGGL(k_if_else, dim3(1), dim3(128), 0, nullptr, 0);
  unsigned long long m3 = 0, odd_rest = 0;
  for (unsigned l = 0; l < 64; l++) {
    if (l % 3 == 0) m3 |= 1ull << l;
    else if (l & 1) odd_rest |= 1ull << l;
  }
  for (unsigned t = 0; t < 128; t++) {
    CHECK(out64[t] == ((t & 63) % 3 == 0 ? m3 : odd_rest));
    CHECK(out64[512 + t] == ~0ull);
  }
  hipLaunchKernelGGL(k_ragged_loop, dim3(2), dim3(64), 0, nullptr, 0);
  for (unsigned l = 0; l < 64; l++) {
    unsigned want = 0;
    for (unsigned i = 0; i < l % 5; i++) {
      unsigned c = 0;
      for (unsigned k = 0; k < 64; k++) c += (k % 5) > i;
      want += c;
    }
    CHECK(out32[l] == want);
    CHECK(out64[l] == ~0ull);
  }
  hipLaunchKernelGGL(k_early_exit, dim3(1), dim3(64), 0, nullptr, 0);
  {
    unsigned long long stay = 0;
    for (unsigned l = 0; l < 64; l++)
      if (l < 8 || !(l & 1)) stay |= 1ull << l;
    unsigned rank = 0;
    for (unsigned l = 0; l < 64; l++) {
      if (l >= 8 && !(l & 1)) {
        CHECK(out32[l] == 100 + rank);
        rank++;
      } else {
        CHECK(out32[l] == 0xffffffffu);
      }
      if (l < 8 || !(l & 1)) CHECK(out64[l] == stay);
    }
  }
  hipLaunchKernelGGL(k_shuffles, dim3(1), dim3(64), 0, nullptr, 0);
  for (unsigned l = 0; l < 64; l++) {
    CHECK(out32[l] == 63);
    CHECK(out32[256 + l] == 63);
    if (l >= 40) CHECK(out32[512 + l] == 40);
  }
  hipLaunchKernelGGL(k_barrier, dim3(3), dim3(256), 0, nullptr, 0);
  for (unsigned t = 0; t < 256; t++) {
    CHECK(out32[t] == (255 - t) * (255 - t));
    CHECK(out32[256 + t] == 7);
  }
  printf(failures ? "FAILED (%d)\n" : "ok\n", failures);
  return failures ? 1 : 0;
}
