// zkw_expand_kernel — the 512-byte CycleRecords of batches, materialised ON THE DEVICE for a consumer that lives there.
//
// The tracer contract of the reference is a full VmLocalState per cycle: start_new_execution_cycle(&local_state) /
// end_execution_cycle(&local_state) (witness_trace/mod.rs:11-20, cycle.rs:34,413).  The cycle kernel stores that
// snapshot losslessly in delta form (a 16-byte tail per cycle + the 32-byte values of the registers the cycle
// wrote + the slow half of the tail when it changes); zkw_batch_get_instance_trace rebuilds the snapshots on the host.  This
// kernel is the same rebuild as a streaming kernel.  One workgroup (4 waves) per (wave of a batch, chunk of cycles): the
// current snapshot of the wave's 64 instances lives in LDS ([lane][32] x 16 B, rows padded to 33 units); per VM cycle the
// first wave applies the cycle's deltas (positions from the masks in the tails, exactly the order zkw_cycle_kernel wrote
// them in) and refreshes the two tail units, then all four waves stream the 64 records out — two whole records
// (2 x 512 contiguous bytes) per store instruction.  A trace is a sequential chain (cycle k + 1 patches the snapshot of
// cycle k), so a launch with few waves is cut into chunks of cycles: a workgroup first replays the cycles in front of its
// chunk silently (tails + deltas only: a tenth of the bytes it then writes), which is what fills the chip from one
// 4096-instance batch; a fused group of batches has waves enough and runs unchunked.
// Bound: HBM writes (512 B out per cycle against ~50 B in).
#include <hip/hip_runtime.h>

#include <cstring>

#include "zkw_device.h"

typedef uint32_t u32;
typedef uint64_t u64;

#define ZKW_EXPAND_ROW 33u /* 16-byte units per LDS row: 32 of the record + 1 of padding (bank spread of the per-lane writes) */
#define ZKW_EXPAND_MAX 128 /* batches per launch (the by-value table stays under the 4 KB kernel-argument segment) */

typedef struct zkw_expand_args {
  const zkw_kparams* kp[ZKW_EXPAND_MAX];
  uint4* dst[ZKW_EXPAND_MAX];        /* record (instance - first, cycle k) of batch b at 16-byte unit ((instance - first) * stride_i + k * stride_k) * 32 of dst[b] */
  u32 wave_base[ZKW_EXPAND_MAX + 1]; /* the waves of the launch numbered through its batches */
  u64 stride_i, stride_k;            /* records between consecutive instances / consecutive cycles in dst */
  u32 n_batches;
  u32 first, count;                  /* instances [first, first + count) of every batch (a ranged call has one batch) */
  u32 first_wave;                    /* first / L of a ranged call */
  u32 chunk_len;                     /* cycles per chunk (grid.y chunks) */
  u32 reserved;
} zkw_expand_args;

struct zkw_expand_lane {  // what a lane carries from cycle to cycle besides its LDS row
  u32 heap_bound, aux_bound, depth, timestamp, pc;
};

// cycle k of the wave: deltas into the LDS rows (+ the lane's slow fields), then timestamp / pc; returns the tail
static __device__ __forceinline__ uint4 zkw_expand_apply(const zkw_kparams ZKW_CONST_AS& P, uint4* snap, const uint4* dl, u32 n_delta, u32 wave, u32 lane, bool live,
                                                         uint4 t0, u32 k, zkw_expand_lane& s) {
  const u32 mask = live ? ((t0.x >> 24) | ((t0.w >> 24) << 8)) : 0u;
  u32 pos = P.dir[((u64)wave * (P.max_cycles + 1) + k) * 4 + 3];
  // order inside a wave-cycle: by mask bit (ascending), lanes in lane order within a bit (zkw_cycle_kernel)
#ifdef __HIP_DEVICE_COMPILE__
  u32 any = mask;  // union of the lanes' masks (wave-uniform)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) any |= (u32)__shfl_xor((int)any, off);
  any = (u32)__builtin_amdgcn_readfirstlane((int)any);
#else
  u32 any = mask;
#endif
  for (u32 left = any; left; left &= left - 1u) {
    const u32 r = (u32)__ffsll((long long)left) - 1u;
    const bool has = (mask >> r) & 1u;
    const unsigned long long part = __ballot(has ? 1 : 0);
    if (has) {
#ifdef __HIP_DEVICE_COMPILE__
      const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(part >> 32), __builtin_amdgcn_mbcnt_lo((u32)part, 0u));
#else
      const u32 rank = 0;
#endif
      const u32 at = pos + rank;
      if (at < n_delta) {
        const uint4 lo = dl[at], hi = dl[(u64)P.cap_delta + at];
        if (r < ZKW_REGISTERS_COUNT) {
          snap[lane * ZKW_EXPAND_ROW + 2 * r] = lo;
          snap[lane * ZKW_EXPAND_ROW + 2 * r + 1] = hi;
        } else {
          s.heap_bound = lo.x; s.aux_bound = lo.y; s.depth = lo.z;
        }
      }
    }
    pos += (u32)__popcll(part);
  }
  return t0;
}

__global__ void __launch_bounds__(256) zkw_expand_kernel(zkw_expand_args X) {
  __shared__ uint4 snap[64 * ZKW_EXPAND_ROW];
  __shared__ u32 ncyc_s[64];
  // batch of this workgroup: largest b with wave_base[b] <= blockIdx.x
  u32 lo = 0, hi = X.n_batches;
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (X.wave_base[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const zkw_kparams ZKW_CONST_AS& P = *(const zkw_kparams ZKW_CONST_AS*)X.kp[lo];
  uint4* const dst = X.dst[lo];
  const u32 L = P.L, tid = threadIdx.x, nthreads = blockDim.x;
  const u32 wave = X.first_wave + (blockIdx.x - X.wave_base[lo]);
  if (wave >= P.n_waves) return;
  const u32 lane = tid;  // (of the first wave: the owners)
  const u32 inst = wave * L + lane;
  const bool owner = tid < L && inst < P.n_instances;
  u32 ncyc = 0;
  zkw_expand_lane s;
  s.heap_bound = s.aux_bound = s.depth = s.timestamp = s.pc = 0;
  if (owner) {
    const zkw_dev_scalars sc0 = P.scalars0[inst];
    ncyc = P.scalars[inst].n_cycles;
    const u32* e = (const u32*)(P.callstack0 + (u64)inst * (P.D + 1) + sc0.depth);
    s.heap_bound = e[26]; s.aux_bound = e[27]; s.depth = sc0.depth; s.timestamp = sc0.timestamp; s.pc = e[17] >> 16;
    for (u32 ch = 0; ch < ZKW_REG_CHUNKS; ch++) snap[lane * ZKW_EXPAND_ROW + ch] = P.regs0[((u64)wave * ZKW_REG_CHUNKS + ch) * L + lane];
  }
  if (tid < 64) ncyc_s[tid] = ncyc;
  __syncthreads();
  u32 n_max = 0;
  for (u32 l = 0; l < L && l < 64; l++) n_max = ncyc_s[l] > n_max ? ncyc_s[l] : n_max;
  const u32 k0 = blockIdx.y * X.chunk_len;
  if (k0 >= n_max) return;  // (uniform: every thread of the workgroup)
  const u32 k1 = k0 + X.chunk_len < n_max ? k0 + X.chunk_len : n_max;
  const u32 cur_delta = P.cursors[wave * 4 + 3];
  const u32 n_delta = cur_delta < P.cap_delta ? cur_delta : P.cap_delta;
  const uint4* dl = P.deltas + (u64)wave * P.cap_delta * 2;
  const u32 time_delta = P.consts.time_delta_per_cycle;
  const uint4* tails = P.tails + (u64)wave * P.max_cycles * L + lane;
  const bool first_wave = tid < 64;  // (emulation build: the one thread)
  // the cycles in front of the chunk, silently (first wave only; the tail of the next cycle is requested before this one's
  // deltas are waited for)
  uint4 t_next = make_uint4(0, 0, 0, 0);
  if (first_wave && owner && 0 < ncyc) t_next = tails[0];
  if (first_wave) {
    for (u32 k = 0; k < k0; k++) {
      const bool live = owner && k < ncyc;
      const uint4 t0 = t_next;
      if (owner && k + 1 < ncyc) t_next = tails[(u64)(k + 1) * L];
      zkw_expand_apply(P, snap, dl, n_delta, wave, lane, live, t0, k, s);
      if (live) {
        s.timestamp += time_delta;
        s.pc = t0.y & 0xffffu;
      }
    }
  }
  for (u32 k = k0; k < k1; k++) {
    if (first_wave) {
      const bool live = owner && k < ncyc;
      const uint4 t0 = t_next;
      if (owner && k + 1 < ncyc) t_next = tails[(u64)(k + 1) * L];
      zkw_expand_apply(P, snap, dl, n_delta, wave, lane, live, t0, k, s);
      if (live) {
        // timestamp and previous_super_pc are not stored: the one advances by a constant per completed cycle (cycle.rs:408-411),
        // the other is the super-pc the cycle started from (cycle.rs:84,113)
        const u32 super_pc = (s.pc & 0xffffu) >> 2;
        s.timestamp += time_delta;
        s.pc = t0.y & 0xffffu;
        snap[lane * ZKW_EXPAND_ROW + 30] = make_uint4(t0.x & 0x00ffffffu, t0.y, t0.z, s.timestamp);
        snap[lane * ZKW_EXPAND_ROW + 31] = make_uint4(s.heap_bound, s.aux_bound, (s.depth & 0xffffu) | (super_pc << 16), t0.w & 0x00ffffffu);
      }
    }
    __syncthreads();
    // stream the snapshots out: thread t handles unit t % 32 of record t / 32 (+ 8 per round of the 256 threads)
    for (u32 idx = tid; idx < L * 32u; idx += nthreads) {
      const u32 rec = idx >> 5, c = idx & 31u;
      const u32 ri = wave * L + rec;
      if (k < ncyc_s[rec] && ri >= X.first && ri - X.first < X.count) {
        uint4* out = dst + ((u64)(ri - X.first) * X.stride_i + (u64)k * X.stride_k) * 32u + c;
#ifdef __HIP_DEVICE_COMPILE__
        typedef unsigned int zkw_v4u __attribute__((ext_vector_type(4)));
        const uint4 v = snap[rec * ZKW_EXPAND_ROW + c];
        zkw_v4u t;
        t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, (zkw_v4u*)out);
#else
        *out = snap[rec * ZKW_EXPAND_ROW + c];
#endif
      }
    }
    __syncthreads();
  }
}

// n batches, instances [first, first + count) of each (a ranged call passes one batch); n_cus: the launch is cut into
// chunks of cycles while it has fewer workgroups than ~4 per CU
extern "C" hipError_t zkw_launch_expand(const zkw_kparams* const* kp, void* const* dst, const uint32_t* n_waves, uint32_t n, uint64_t stride_i, uint64_t stride_k, uint32_t first,
                                        uint32_t count, uint32_t L, uint32_t wave_threads, uint32_t max_cycles_run, uint32_t n_cus, hipStream_t stream) {
  zkw_expand_args X;
  memset(&X, 0, sizeof X);
  X.n_batches = n;
  for (uint32_t i = 0; i < n; i++) {
    X.kp[i] = kp[i];
    X.dst[i] = (uint4*)dst[i];
    X.wave_base[i + 1] = X.wave_base[i] + n_waves[i];
  }
  X.stride_i = stride_i; X.stride_k = stride_k; X.first = first; X.count = count;
  const bool ranged = n == 1 && count != 0xffffffffu;  // (a ranged call: the waves that hold instances [first, first + count) of the one batch)
  X.first_wave = ranged ? first / L : 0;
  if (ranged) X.wave_base[1] = (first + count - 1) / L - X.first_wave + 1;
  const uint32_t waves = X.wave_base[n];
  uint32_t chunks = 1;
  if (wave_threads > 1 && max_cycles_run > 1) {
    while (waves * chunks < 4u * n_cus && chunks * 2u <= 16u && max_cycles_run / (chunks * 2u) >= 8u) chunks *= 2u;
  }
  X.chunk_len = (max_cycles_run + chunks - 1) / chunks;
  if (X.chunk_len == 0) X.chunk_len = 1;
  hipLaunchKernelGGL(zkw_expand_kernel, dim3(waves, chunks), dim3(wave_threads > 1 ? 256 : 1), 0, stream, X);
  return hipGetLastError();
}
