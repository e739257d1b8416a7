// zkw_expand_kernel — the 512-byte CycleRecords of a batch, materialised ON THE DEVICE for a consumer that lives there.
//
// The tracer contract of the reference is a full VmLocalState per cycle: start_new_execution_cycle(&local_state) /
// end_execution_cycle(&local_state) (witness_trace/mod.rs:11-20, cycle.rs:34,413).  The cycle kernel stores that
// snapshot losslessly in delta form (a 16-byte tail per cycle + the 32-byte values of the registers the cycle
// wrote + the slow half of the tail when it changes); zkw_batch_get_instance_trace rebuilds the snapshots on the host.  This
// kernel is the same rebuild as a streaming kernel: one wave per wave of the batch, the current snapshot of its 64
// instances in LDS ([lane][32] x 16 B, rows padded to 33 units), per VM cycle: apply the cycle's deltas (positions from the
// masks in the tails, exactly the order zkw_cycle_kernel wrote them in), refresh the two tail units, and stream the 64
// records out — two whole records (2 x 512 contiguous bytes) per store instruction.
// Bound: HBM writes (512 B out per cycle against ~50 B in).
#include <hip/hip_runtime.h>

#include "zkw_device.h"

typedef uint32_t u32;
typedef uint64_t u64;

#define ZKW_EXPAND_ROW 33u /* 16-byte units per LDS row: 32 of the record + 1 of padding (bank spread of the per-lane writes) */

typedef struct zkw_expand_params {
  const zkw_kparams* kp;
  const zkw_dev_entry* callstack0;  /* pristine callstack: the pc / memory bounds an instance started with */
  uint4* dst;                       /* record (instance - first, cycle k) at 16-byte unit ((instance - first) * stride + k) * 32 */
  u64 stride;                       /* records per instance in dst */
  u32 first, count;                 /* instances [first, first + count) */
  u32 first_wave;
  u32 reserved;
} zkw_expand_params;

__global__ void zkw_expand_kernel(zkw_expand_params X) {
  const zkw_kparams ZKW_CONST_AS& P = *(const zkw_kparams ZKW_CONST_AS*)X.kp;
  __shared__ uint4 snap[64 * ZKW_EXPAND_ROW];
  __shared__ u32 ncyc_s[64];
  const u32 L = P.L, lane = threadIdx.x, nthreads = blockDim.x;
  const u32 wave = X.first_wave + blockIdx.x;
  if (wave >= P.n_waves) return;
  const u32 inst = wave * L + lane;
  const bool owner = lane < L && inst < P.n_instances;
  u32 ncyc = 0, heap_bound = 0, aux_bound = 0, depth = 0, timestamp = 0, pc = 0;
  if (owner) {
    const zkw_dev_scalars sc = P.scalars[inst], sc0 = P.scalars0[inst];
    ncyc = sc.n_cycles;
    const u32* e = (const u32*)(X.callstack0 + (u64)inst * (P.D + 1) + sc0.depth);
    heap_bound = e[26]; aux_bound = e[27]; depth = sc0.depth; timestamp = sc0.timestamp; pc = e[17] >> 16;
    for (u32 ch = 0; ch < ZKW_REG_CHUNKS; ch++) snap[lane * ZKW_EXPAND_ROW + ch] = P.regs0[((u64)wave * ZKW_REG_CHUNKS + ch) * L + lane];
  }
  if (lane < 64) ncyc_s[lane] = ncyc;
  __syncthreads();
  u32 n_max = 0;
  for (u32 l = 0; l < L && l < 64; l++) n_max = ncyc_s[l] > n_max ? ncyc_s[l] : n_max;
  const u32 cur_delta = P.cursors[wave * 4 + 3];
  const u32 n_delta = cur_delta < P.cap_delta ? cur_delta : P.cap_delta;
  const uint4* dl = P.deltas + (u64)wave * P.cap_delta * 2;
  const u32 time_delta = P.consts.time_delta_per_cycle;
  for (u32 k = 0; k < n_max; k++) {
    const bool live = owner && k < ncyc;
    uint4 t0 = make_uint4(0, 0, 0, 0);
    if (live) t0 = P.tails[((u64)wave * P.max_cycles + k) * L + lane];
    const u32 mask = live ? ((t0.x >> 24) | ((t0.w >> 24) << 8)) : 0u;
    u32 pos = P.dir[((u64)wave * (P.max_cycles + 1) + k) * 4 + 3];
    // order inside a wave-cycle: by mask bit (ascending), lanes in lane order within a bit (zkw_cycle_kernel)
    for (u32 r = 0; r < ZKW_REGISTERS_COUNT + 1; r++) {
      const bool has = (mask >> r) & 1u;
      const unsigned long long part = __ballot(has ? 1 : 0);
      if (!part) continue;
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
            heap_bound = lo.x; aux_bound = lo.y; depth = lo.z;
          }
        }
      }
      pos += (u32)__popcll(part);
    }
    if (live) {
      // timestamp and previous_super_pc are not stored: the one advances by a constant per completed cycle (cycle.rs:408-411),
      // the other is the super-pc the cycle started from (cycle.rs:84,113)
      const u32 super_pc = (pc & 0xffffu) >> 2;
      timestamp += time_delta;
      pc = t0.y & 0xffffu;
      snap[lane * ZKW_EXPAND_ROW + 30] = make_uint4(t0.x & 0x00ffffffu, t0.y, t0.z, timestamp);
      snap[lane * ZKW_EXPAND_ROW + 31] = make_uint4(heap_bound, aux_bound, (depth & 0xffffu) | (super_pc << 16), t0.w & 0x00ffffffu);
    }
    __syncthreads();
    // stream the snapshots out: thread t handles unit t % 32 of record 2 j + t / 32
    for (u32 idx = lane; idx < L * 32u; idx += nthreads) {
      const u32 rec = idx >> 5, c = idx & 31u;
      const u32 ri = wave * L + rec;
      if (k < ncyc_s[rec] && ri >= X.first && ri < X.first + X.count) {
        uint4* out = X.dst + ((u64)(ri - X.first) * X.stride + k) * 32u + c;
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

extern "C" hipError_t zkw_launch_expand(const zkw_kparams* kp, const zkw_dev_entry* callstack0, void* dst, uint64_t stride, uint32_t first, uint32_t count,
                                        uint32_t L, uint32_t wave_threads, hipStream_t stream) {
  zkw_expand_params X;
  X.kp = kp; X.callstack0 = callstack0; X.dst = (uint4*)dst; X.stride = stride; X.first = first; X.count = count;
  X.first_wave = first / L;
  X.reserved = 0;
  const uint32_t last_wave = (first + count - 1) / L;
  hipLaunchKernelGGL(zkw_expand_kernel, dim3(last_wave - X.first_wave + 1), dim3(wave_threads), 0, stream, X);
  return hipGetLastError();
}
