// TEST INFRASTRUCTURE ONLY — never part of the product, never loaded by the package.
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile era-zk_evm_amd/csrc/*.hip|cpp
// UNMODIFIED into tests/emu/libzkw_emu*.so, so that the `-m "not gpu"` suite can run the real kernel
// logic and host runtime (decode, opcodes, stream compaction indices, gather) against the oracle
// on a box without a GPU.  Two flavours (build_emu.py):
//   ZKW_EMU_WAVE == 1  (libzkw_emu.so)   a single-lane "wave" (warpSize 1) executed sequentially: fast, no lane-parallel
//                                        behaviour;
//   ZKW_EMU_WAVE == 64 (libzkw_emu64.so) 64-lane waves on the SIMT engine of emu_simt.cpp: every thread of a workgroup
//                                        is a fiber, cross-lane operations and barriers are rendezvous points, the
//                                        execution mask comes from the ZKW_DIV_IF / ZKW_DIV_SCOPE annotations of the
//                                        kernel source — ballot ranks, group selection, the short cycle's wave-uniform
//                                        tests and divergent lanes run on the CPU.
// (LDS banking, coalescing and timing are only ever seen by the `-m gpu` tests on a real MI355X.)
#pragma once
#define ZKW_EMU_BUILD 1 /* the product sources see this only in the tests/emu build */
#ifndef ZKW_EMU_WAVE
#define ZKW_EMU_WAVE 1
#endif
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <chrono>

#define __global__
#define __device__
#define __host__
#if ZKW_EMU_WAVE > 1
#define __shared__ static /* one workgroup runs at a time (emu_simt.cpp): a static is what its fibers share */
#else
#define __shared__
#endif
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;
extern uint4 zkw_lds[];  // the dynamic LDS segment

static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
// cross-lane operations (ZKW_EMU_WAVE > 1: rendezvous points of the lanes' fibers, emu_simt.cpp)
enum { ZE_BALLOT, ZE_READFIRST, ZE_READLANE, ZE_SHFL, ZE_SHFL_XOR, ZE_BPERMUTE, ZE_WAVE_BARRIER, ZE_FETCH_ADD, ZE_YIELD, ZE_SCOPE_BEGIN, ZE_SYNCTHREADS };
#if ZKW_EMU_WAVE > 1
extern "C" uint64_t zkw_emu_collective(int kind, uint64_t operand, uint32_t arg, const char* file, int line);
extern "C" void zkw_emu_scope_end(void);
extern "C" uint32_t* zkw_emu_wave_sregs(void);
extern "C" void zkw_emu_launch(void (*entry)(void*), void* arg, dim3 grid, dim3 block);
#define __ballot(p) ((unsigned long long)zkw_emu_collective(ZE_BALLOT, (p) ? 1u : 0u, 0u, __FILE__, __LINE__))
#define __shfl(v, l) ((int)zkw_emu_collective(ZE_SHFL, (uint32_t)(v), (uint32_t)(l), __FILE__, __LINE__))
#define __shfl_xor(v, o) ((int)zkw_emu_collective(ZE_SHFL_XOR, (uint32_t)(v), (uint32_t)(o), __FILE__, __LINE__))
#define __syncthreads() ((void)zkw_emu_collective(ZE_SYNCTHREADS, 0u, 0u, __FILE__, __LINE__))
// A divergence scope = what the execution mask does on the hardware: the lanes whose condition holds run the region
// while the others wait, then the others run (the else arm, or nothing), and all of them continue together behind it —
// also the lanes that left the region early (break / continue / return run the destructor).
struct zkw_emu_scope {
  bool taken;
  zkw_emu_scope(bool c, const char* f, int l) { taken = zkw_emu_collective(ZE_SCOPE_BEGIN, c ? 1u : 0u, 0u, f, l) != 0; }
  ~zkw_emu_scope() { zkw_emu_scope_end(); }
  zkw_emu_scope(const zkw_emu_scope&) = delete;
};
#define ZKW_DIV_IF(c) if (zkw_emu_scope zkw_sc{(bool)(c), __FILE__, __LINE__}; zkw_sc.taken)
#define ZKW_DIV_SCOPE zkw_emu_scope zkw_sc_region{true, __FILE__, __LINE__}
// "the lanes of the wave are in lockstep here": between a read of wave-shared memory by every lane and its update by one
#define ZKW_LOCKSTEP() ((void)zkw_emu_collective(ZE_WAVE_BARRIER, 0u, 0u, __FILE__, __LINE__))
static inline uint32_t zkw_emu_fetch_add(uint32_t reg, uint32_t n, const char* f, int l) { return (uint32_t)zkw_emu_collective(ZE_FETCH_ADD, n, reg, f, l); }
#define ZKW_EMU_FETCH_ADD(reg, n) zkw_emu_fetch_add((reg), (n), __FILE__, __LINE__)
#define ZKW_EMU_YIELD() ((void)zkw_emu_collective(ZE_YIELD, 0u, 0u, __FILE__, __LINE__))
#else
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
static inline int __shfl(int v, int) { return v; }
static inline int __shfl_xor(int v, int) { return v; }
static inline void __syncthreads() {}
#define ZKW_DIV_IF(c) if (c)
#define ZKW_DIV_SCOPE ((void)0)
#define ZKW_LOCKSTEP() ((void)0)
extern uint32_t zkw_emu_sregs_1[64][16];  // one-lane waves: the "wave" is the thread
static inline uint32_t* zkw_emu_wave_sregs(void) { return zkw_emu_sregs_1[threadIdx.x & 63u]; }
static inline uint32_t zkw_emu_fetch_add_1(uint32_t reg, uint32_t n) { uint32_t* r = zkw_emu_wave_sregs(); const uint32_t o = r[reg]; r[reg] = o + n; return o; }
#define ZKW_EMU_FETCH_ADD(reg, n) zkw_emu_fetch_add_1((reg), (n))
#define ZKW_EMU_YIELD() ((void)0)
#endif
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotSupported = 801 };
typedef void* hipStream_t;
struct EmuEvent { std::chrono::steady_clock::time_point t; };
typedef EmuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount, hipDeviceAttributeWarpSize };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeWarpSize ? ZKW_EMU_WAVE : 4; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; r++) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}

#if ZKW_EMU_WAVE > 1
template <typename K, typename... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
  auto thunk = [&]() { kernel(args...); };
  typedef decltype(thunk) thunk_t;
  zkw_emu_launch([](void* p) { (*static_cast<thunk_t*>(p))(); }, &thunk, grid, block);
}
#else
template <typename K, typename... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
  gridDim = grid;
  blockDim = block;
  for (unsigned bz = 0; bz < grid.z; bz++)
  for (unsigned by = 0; by < grid.y; by++)
    for (unsigned b = 0; b < grid.x; b++) {
      blockIdx = dim3(b, by, bz);
      for (unsigned t = 0; t < block.x; t++) {  // block.x == 1 (warpSize 1)
        threadIdx = dim3(t);
        kernel(args...);
      }
    }
}
#endif

// wave-uniform broadcast of the first active lane: identity for a one-lane wave
#if ZKW_EMU_WAVE > 1
#define __builtin_amdgcn_readfirstlane(x) ((int)zkw_emu_collective(ZE_READFIRST, (uint32_t)(x), 0u, __FILE__, __LINE__))
#define __builtin_amdgcn_readlane(x, l) ((int)zkw_emu_collective(ZE_READLANE, (uint32_t)(x), (uint32_t)(l), __FILE__, __LINE__))
#define __builtin_amdgcn_ds_bpermute(a, v) ((int)zkw_emu_collective(ZE_BPERMUTE, (uint32_t)(v), (uint32_t)(a), __FILE__, __LINE__))
#else
static inline int zkw_emu_readfirstlane(int x) { return x; }
#define __builtin_amdgcn_readfirstlane(x) zkw_emu_readfirstlane(x)
static inline int zkw_emu_readlane(int x, int) { return x; }
#define __builtin_amdgcn_readlane(x, l) zkw_emu_readlane(x, l)
#endif
static inline uint32_t zkw_emu_alignbit(uint32_t hi, uint32_t lo, uint32_t n) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (n & 31u)); }
#define __builtin_amdgcn_alignbit(hi, lo, n) zkw_emu_alignbit(hi, lo, n)

// stream capture / graphs: not emulated — zkw_batch_step falls back to its eager sequence
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

#define __builtin_amdgcn_fence(order, scope) ((void)0)
#if ZKW_EMU_WAVE > 1
#define __builtin_amdgcn_wave_barrier() ((void)zkw_emu_collective(ZE_WAVE_BARRIER, 0u, 0u, __FILE__, __LINE__))
#else
#define __builtin_amdgcn_wave_barrier() ((void)0)
#endif
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

#define __noinline__ __attribute__((noinline))

enum { hipHostMallocDefault = 0 };
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }

// round 5 (zkw_delivery / zkw_batch_restage): streams of their own, stream-to-event ordering, device view of pinned memory —
// all synchronous here
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
enum { hipEventBlockingSync = 1 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
