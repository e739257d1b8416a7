// zkw_expand_kernel — the 512-byte CycleRecords of batches, materialised ON THE DEVICE for a consumer that lives there.
//
// The tracer contract of the reference is a full VmLocalState per cycle: start_new_execution_cycle(&local_state) /
// end_execution_cycle(&local_state) (witness_trace/mod.rs:11-20, cycle.rs:34,413).  The cycle kernel stores that
// snapshot losslessly in delta form (a 16-byte tail per cycle + the 32-byte values of the registers the cycle
// wrote + the slow half of the tail when it changes); zkw_batch_get_instance_trace rebuilds the snapshots on the host.  This
// kernel is the same rebuild as a streaming kernel.  One workgroup (4 waves) per (wave of a batch, chunk of cycles): the
// current snapshot of the wave's 64 instances lives in LDS ([lane][32] x 16 B = 32 KB exactly: FIVE workgroups per CU, so the
// 1280 waves of the driver's 20 batches are one round of the chip; unit c of a row sits at c ^ (lane & 31), which spreads the
// per-lane writes over the banks the way a padded row did and keeps a row's 32 units one contiguous 512 bytes); per VM cycle the
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

#define ZKW_EXPAND_ROW 32u /* 16-byte units per LDS row = one record; no padding: 64 rows are 32 KB, five workgroups fill the 160 KB of a CU */
// unit `c` of row `row` (the swizzle spreads the 64 lanes that write the same unit of their rows over the banks)
#define ZKW_EXPAND_AT(row, c) ((row) * ZKW_EXPAND_ROW + ((c) ^ ((row) & 31u)))
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

// rank of this lane among the set bits of a wave mask
#ifdef __HIP_DEVICE_COMPILE__
#define ZKW_X_RANK(part) __builtin_amdgcn_mbcnt_hi((u32)((part) >> 32), __builtin_amdgcn_mbcnt_lo((u32)(part), 0u))
#else
#define ZKW_X_RANK(part) ((u32)__popcll((part) & ((1ull << (threadIdx.x & 63u)) - 1ull)))
#endif

struct zkw_expand_lane {  // what a lane carries from cycle to cycle besides its LDS row
  u32 heap_bound, aux_bound, depth, timestamp, pc;
};

// cycle k of the wave: deltas into the LDS rows (+ the lane's slow fields); `pos` = the wave's delta cursor at the start of
// the cycle (directory entry k, word 3).  (The emulation build's form; on the device the applying wave pipelines the same steps.)
[[maybe_unused]] static __device__ __forceinline__ void zkw_expand_apply(const zkw_kparams ZKW_CONST_AS& P, uint4* snap, const uint4* dl, u32 n_delta, u32 lane, bool live, uint4 t0, u32 pos,
                                                        zkw_expand_lane& s) {
  const u32 mask = live ? ((t0.x >> 24) | ((t0.w >> 24) << 8)) : 0u;
  // order inside a wave-cycle: by mask bit (ascending), lanes in lane order within a bit (zkw_cycle_kernel)
#ifdef __HIP_DEVICE_COMPILE__
  // union of the lanes' masks (wave-uniform).  A shared tape makes all masks equal: one compare; else one ballot per bit
  const u32 m0 = (u32)__builtin_amdgcn_readfirstlane((int)mask);
  u32 any = m0;
  if (__ballot(mask != m0 ? 1 : 0) != 0) {
    any = 0;
#pragma unroll
    for (u32 r = 0; r < ZKW_REGISTERS_COUNT + 1; r++) any |= __ballot((mask >> r) & 1u) ? 1u << r : 0u;
  }
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
          snap[ZKW_EXPAND_AT(lane, 2 * r)] = lo;
          snap[ZKW_EXPAND_AT(lane, 2 * r + 1)] = hi;
        } else {
          s.heap_bound = lo.x; s.aux_bound = lo.y; s.depth = lo.z;
        }
      }
    }
    pos += (u32)__popcll(part);
  }
}

// the two tail units of a lane's row after cycle k.  timestamp and previous_super_pc are not stored: the one advances by a
// constant per completed cycle (cycle.rs:408-411), the other is the super-pc the cycle started from (cycle.rs:84,113)
static __device__ __forceinline__ void zkw_expand_tail(uint4* snap, u32 lane, uint4 t0, u32 time_delta, zkw_expand_lane& s) {
  const u32 super_pc = (s.pc & 0xffffu) >> 2;
  s.timestamp += time_delta;
  s.pc = t0.y & 0xffffu;
  snap[ZKW_EXPAND_AT(lane, 30u)] = make_uint4(t0.x & 0x00ffffffu, t0.y, t0.z, s.timestamp);
  snap[ZKW_EXPAND_AT(lane, 31u)] = make_uint4(s.heap_bound, s.aux_bound, (s.depth & 0xffffu) | (super_pc << 16), t0.w & 0x00ffffffu);
}

// threads of a workgroup on the device: one wave that applies (the owners of the 64 rows) + four that stream
#define ZKW_EXPAND_THREADS 320
__global__ void __launch_bounds__(ZKW_EXPAND_THREADS, 7) zkw_expand_kernel(zkw_expand_args X) {
  // (the rows are ALL of the workgroup's LDS: the cycle counts of the 64 instances pass through the rows once, before the
  // snapshots are loaded, and then live in registers — 256 more bytes would cost the fifth workgroup of a CU)
  __shared__ uint4 snap[64 * ZKW_EXPAND_ROW];
  // batch of this workgroup: largest b with wave_base[b] <= blockIdx.x
  u32 lo = 0, hi = X.n_batches;
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (X.wave_base[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  const zkw_kparams ZKW_CONST_AS& P = *(const zkw_kparams ZKW_CONST_AS*)X.kp[lo];
  uint4* const dst = X.dst[lo];
  const u32 L = P.L, tid = threadIdx.x;
  const u32 wave = X.first_wave + (blockIdx.x - X.wave_base[lo]);
  if (wave >= P.n_waves) return;
  // Roles.  The first wave owns the rows: lane = instance, it applies the deltas of a cycle and never stores — its loads
  // would queue behind its own stream stores otherwise (vmcnt retires in order, and a store is acknowledged microseconds
  // later when the chip is writing at its ceiling: the wave that did both made the kernel 30 % slower than its stores
  // alone).  The other four waves stream the 64 records out.  (Emulation build: the one thread does both.)
  const bool first_wave = tid < 64;
  const u32 lane = tid;
  const u32 inst = wave * L + lane;
  const bool owner = first_wave && tid < L && inst < P.n_instances;
  u32 ncyc = 0;
  zkw_expand_lane s;
  s.heap_bound = s.aux_bound = s.depth = s.timestamp = s.pc = 0;
  if (owner) {
    const zkw_dev_scalars sc0 = P.scalars0[inst];
    ncyc = P.scalars[inst].n_cycles;
    const u32* e = (const u32*)(P.callstack0 + (u64)inst * (P.D + 1) + sc0.depth);
    s.heap_bound = e[26]; s.aux_bound = e[27]; s.depth = sc0.depth; s.timestamp = sc0.timestamp; s.pc = e[17] >> 16;
  }
  u32* const ncyc_s = (u32*)snap;
  if (tid < 64) ncyc_s[tid] = ncyc;
  __syncthreads();
  u32 n_max = 0;
  for (u32 l = 0; l < L && l < 64; l++) n_max = ncyc_s[l] > n_max ? ncyc_s[l] : n_max;
  // the records a streaming thread writes: unit st % 32 of records st / 32 + 8 j (st = 0..255).  It keeps two numbers: up to
  // cycle `kmin` all eight of them exist (and are inside the requested range), from `kmax` on none does; in between — ragged
  // ends, a ranged call — the cycle counts are read again (eight registers more would spill in the stream loop)
#ifdef ZKW_WIDE
  const u32 st = tid - 64u;  // (streaming threads: tid >= 64)
  u32 kmin = 0xffffffffu, kmax = 0;
#pragma unroll
  for (u32 j = 0; j < 8; j++) {
    const u32 rec = (st >> 5) + 8u * j, ri = wave * L + rec;
    const u32 n = (!first_wave && rec < L && ri >= X.first && ri - X.first < X.count) ? ncyc_s[rec] : 0u;
    kmin = n < kmin ? n : kmin;
    kmax = n > kmax ? n : kmax;
  }
#else
  u32 ncyc_e[64];  // (one-lane emulation build: the one thread does both)
  for (u32 l = 0; l < 64; l++) ncyc_e[l] = ncyc_s[l];
#endif
  __syncthreads();
  const u32 k0 = blockIdx.y * X.chunk_len;
  if (k0 >= n_max) return;  // (uniform: every thread of the workgroup)
  if (owner)
    for (u32 ch = 0; ch < ZKW_REG_CHUNKS; ch++) snap[ZKW_EXPAND_AT(lane, ch)] = P.regs0[((u64)wave * ZKW_REG_CHUNKS + ch) * L + lane];
  const u32 k1 = k0 + X.chunk_len < n_max ? k0 + X.chunk_len : n_max;
  const u32 cur_delta = P.cursors[wave * 4 + 3];
  const u32 n_delta = cur_delta < P.cap_delta ? cur_delta : P.cap_delta;
  const uint4* dl = P.deltas + (u64)wave * P.cap_delta * 2;
  const u32 time_delta = P.consts.time_delta_per_cycle;
  const uint4* tails = P.tails + (u64)wave * P.max_cycles * L + lane;
  const u32* dirw = P.dir + (u64)wave * (P.max_cycles + 1) * 4 + 3;  // word 3 of entry k: the delta cursor at the start of wave-cycle k
#ifdef ZKW_WIDE
  if (first_wave) {
    // The applying wave is a software pipeline.  A load of this kernel returns microseconds later (the chip is writing at its
    // ceiling, reads queue behind the writes), and a cycle's delta positions need its tail (the mask) and its directory word:
    // tails and directory words are requested TWO cycles ahead, the deltas of the lowest mask bit of a cycle (a shared tape
    // writes one register per cycle) ONE cycle ahead; further bits are loaded when the cycle is applied.  Every load of the
    // pipeline is unconditional — indices clamped into the arrays, the values of dead lanes / cycles never used — so that the
    // waits are exact counts in straight-line code.
    const u32 last_k = P.max_cycles - 1u;
    const uint4* const tl = owner ? tails : tails - lane;  // (a lane without an instance reads lane 0's tails: never used)
    const u32 nd1 = n_delta ? n_delta - 1u : 0u;
#define ZKW_X_TAIL(k) (tl[(u64)((k) < last_k ? (k) : last_k) * L])
#define ZKW_X_POS(k) (dirw[(u64)((k) <= last_k ? (k) : last_k + 1u) * 4])
#define ZKW_X_MASK(t, k) ((owner && (k) < ncyc) ? (((t).x >> 24) | (((t).w >> 24) << 8)) : 0u)
    // union of the lanes' masks (wave-uniform): a shared tape makes them equal (one compare), else one ballot per bit
    auto any_of = [](u32 mask) -> u32 {
      const u32 m0 = (u32)__builtin_amdgcn_readfirstlane((int)mask);
      u32 any = m0;
      if (__ballot(mask != m0 ? 1 : 0) != 0) {
        any = 0;
#pragma unroll
        for (u32 r = 0; r < ZKW_REGISTERS_COUNT + 1; r++) any |= __ballot((mask >> r) & 1u) ? 1u << r : 0u;
      }
      return any;
    };
    // position of a lane's delta for the lowest bit of `any` in a cycle whose cursor starts at `pos` (clamped into the array)
    auto first_at = [&](u32 mask, u32 any, u32 pos) -> u32 {
      const u32 r = any ? (u32)__ffs((int)any) - 1u : 0u;
      const unsigned long long part = __ballot((mask >> r) & 1u);
      const u32 at = pos + ZKW_X_RANK(part);
      return at < nd1 ? at : nd1;
    };
    // Pipeline state at the top of cycle k: (tc, pc) tail and cursor of cycle k; (clo, chi) the first deltas of cycle k;
    // (q, pq) tail and cursor of cycle k + 1, requested while cycle k - 1 was applied.  Per cycle: from q the positions of
    // the first deltas of cycle k + 1 -> request them; keep q aside and request tail / cursor k + 2 into it; apply cycle k;
    // at the very end move the deltas of k + 1 into place.  Every load has a whole cycle of the workgroup between request and
    // use, and the only registers copied while a load may still be writing them are copied at the end of that cycle.
    uint4 tc = ZKW_X_TAIL(0u);
    u32 pc = ZKW_X_POS(0u);
    uint4 clo, chi;
    {
      const u32 m = ZKW_X_MASK(tc, 0u);
      const u32 at = first_at(m, any_of(m), pc);
      clo = dl[at]; chi = dl[(u64)P.cap_delta + at];
    }
    uint4 q = ZKW_X_TAIL(1u);
    u32 pq = ZKW_X_POS(1u);
    for (u32 k = 0; k < k1; k++) {
      const u32 mask1 = ZKW_X_MASK(q, k + 1u);
      const u32 at1 = first_at(mask1, any_of(mask1), pq);
      const uint4 nlo = dl[at1], nhi = dl[(u64)P.cap_delta + at1];
      const uint4 t1 = q;
      const u32 p1 = pq;
      q = ZKW_X_TAIL(k + 2u);
      pq = ZKW_X_POS(k + 2u);
      // cycle k
      const bool live = owner && k < ncyc;
      const u32 mask = ZKW_X_MASK(tc, k);
#ifndef ZKW_EXPAND_NOAPPLY
      u32 left = any_of(mask), pos = pc;
      bool first = true;  // (wave-uniform)
      // order inside a wave-cycle: by mask bit (ascending), lanes in lane order within a bit (zkw_cycle_kernel)
      for (; left; left &= left - 1u) {
        const u32 r = (u32)__ffs((int)left) - 1u;
        const bool has = (mask >> r) & 1u;
        const unsigned long long part = __ballot(has ? 1 : 0);
        if (has) {
          const u32 at = pos + ZKW_X_RANK(part);
          if (at < n_delta) {
            uint4 lo = clo, hi = chi;
            if (!first) { lo = dl[at]; hi = dl[(u64)P.cap_delta + at]; }
            if (r < ZKW_REGISTERS_COUNT) {
              snap[ZKW_EXPAND_AT(lane, 2 * r)] = lo;
              snap[ZKW_EXPAND_AT(lane, 2 * r + 1)] = hi;
            } else {
              s.heap_bound = lo.x; s.aux_bound = lo.y; s.depth = lo.z;
            }
          }
        }
        pos += (u32)__popcll(part);
        first = false;
      }
#endif
      if (k < k0) {  // in front of the chunk: silently
        if (live) {
          s.timestamp += time_delta;
          s.pc = tc.y & 0xffffu;
        }
      } else {
        if (live) zkw_expand_tail(snap, lane, tc, time_delta, s);
        __syncthreads();
        __syncthreads();
      }
      tc = t1; pc = p1; clo = nlo; chi = nhi;
    }
#undef ZKW_X_TAIL
#undef ZKW_X_POS
#undef ZKW_X_MASK
  } else {
    // stream the snapshots out: two whole records (2 x 512 contiguous bytes) per store instruction of a wave
#ifdef __HIP_DEVICE_COMPILE__
    typedef unsigned int zkw_v4u __attribute__((ext_vector_type(4)));
#endif
    const u32 rec0 = st >> 5, c = st & 31u;
    // (a thread's first record may lie in front of the requested range while its later ones are inside: the difference is signed)
    uint4* const out0 = dst + ((long long)(wave * L + rec0) - (long long)X.first) * (long long)(X.stride_i * 32u) + c;
    const u64 step_j = 8u * X.stride_i * 32u, step_k = X.stride_k * 32u;
    for (u32 k = k0; k < k1; k++) {
      u32 m = k < kmin ? 0xffu : 0u;  // bit j: record rec0 + 8 j has a cycle k (and is wanted)
      if (k >= kmin && k < kmax) {
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
          u32 rec = rec0 + 8u * j;
#ifdef __HIP_DEVICE_COMPILE__
          asm volatile("" : "+v"(rec));  // (opaque: or the eight addresses of this rare path are computed in front of the loop and spilled)
#endif
          const u32 ri = wave * L + rec;
          if (rec < L && ri < P.n_instances && ri >= X.first && ri - X.first < X.count && k < P.scalars[ri].n_cycles) m |= 1u << j;
        }
      }
      __syncthreads();
      uint4 v[8];
#pragma unroll
      for (u32 j = 0; j < 8; j++)
        if ((m >> j) & 1u) v[j] = snap[ZKW_EXPAND_AT(rec0 + 8u * j, c)];
      __syncthreads();
      uint4* const outk = out0 + (u64)k * step_k;
#pragma unroll
      for (u32 j = 0; j < 8; j++) {
        if ((m >> j) & 1u) {
#ifdef __HIP_DEVICE_COMPILE__
          zkw_v4u t;
          t.x = v[j].x; t.y = v[j].y; t.z = v[j].z; t.w = v[j].w;
          __builtin_nontemporal_store(t, (zkw_v4u*)(outk + j * step_j));
#else
          outk[j * step_j] = v[j];
#endif
        }
      }
    }
  }
#else
  // (single-thread emulation build: the same rebuild, one cycle at a time)
  uint4 t_next = make_uint4(0, 0, 0, 0);
  if (owner && 0 < ncyc) t_next = tails[0];
  for (u32 k = 0; k < k1; k++) {
    const bool live = owner && k < ncyc;
    const uint4 t0 = t_next;
    if (owner && k + 1 < ncyc) t_next = tails[(u64)(k + 1) * L];
    zkw_expand_apply(P, snap, dl, n_delta, lane, live, t0, dirw[(u64)k * 4], s);
    if (k < k0) {
      if (live) {
        s.timestamp += time_delta;
        s.pc = t0.y & 0xffffu;
      }
      continue;
    }
    if (live) zkw_expand_tail(snap, lane, t0, time_delta, s);
    for (u32 idx = 0; idx < L * 32u; idx++) {
      const u32 rec = idx >> 5, cc = idx & 31u;
      const u32 ri = wave * L + rec;
      if (k < ncyc_e[rec] && ri >= X.first && ri - X.first < X.count) {
        uint4* out = dst + ((u64)(ri - X.first) * X.stride_i + (u64)k * X.stride_k) * 32u + cc;
        *out = snap[ZKW_EXPAND_AT(rec, cc)];
      }
    }
  }
#endif
}

// n batches, instances [first, first + count) of each (a ranged call passes one batch); n_cus: the launch is cut into
// chunks of cycles while it has fewer workgroups than 5 per CU (what its LDS admits)
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
    while (waves * chunks < 5u * n_cus && chunks * 2u <= 16u && max_cycles_run / (chunks * 2u) >= 8u) chunks *= 2u;
  }
  X.chunk_len = (max_cycles_run + chunks - 1) / chunks;
  if (X.chunk_len == 0) X.chunk_len = 1;
  hipLaunchKernelGGL(zkw_expand_kernel, dim3(waves, chunks), dim3(wave_threads > 1 ? ZKW_EXPAND_THREADS : 1), 0, stream, X);
  return hipGetLastError();
}
