// zkw_pack_kernel — one step's witness, packed for the host (link format: zkw_pack.h).
//
// The consumers the reference names are host code: VmWitnessTracer's ten callbacks (witness_trace/mod.rs:11-72) run on the
// CPU, so what the cycle kernel leaves in HBM has to cross PCIe, and the link (57 GB/s on these boxes) is 100x slower than
// the kernel that fills it.  This kernel is the one consumer of the device streams on that path: it gathers the USED extents
// of every wave of a fused group of batches — directory, record tails, register deltas, the three query streams, final
// scalars / register files / callstack entries — into one contiguous block in a denser encoding, and its destination is the
// pinned host ring itself: the stores ARE the transfer (no staging buffer, no hipMemcpy per stream and wave — round 4 issued
// 576 of them per batch and reached 16 GB/s).
// A persistent grid of a few dozen workgroups (the link, not the chip, is the bound: 64 workgroups saturate it) walks the
// waves; per wave: count (non-code memory queries, used aux units) -> one atomic allocation of the wave's extent in the
// block -> copies.  Bound: PCIe.  Runs beside the cycle kernel of the next group on a stream of its own.
#include <hip/hip_runtime.h>

#include <cstring>

#include "zkw_pack.h"

typedef uint32_t u32;
typedef uint64_t u64;

#ifdef ZKW_WIDE
#define ZKW_PACK_LANE(t) ((t) & 63u)
#define ZKW_PACK_WAVE_OF(t) ((t) >> 6)
#else
#define ZKW_PACK_LANE(t) 0u
#define ZKW_PACK_WAVE_OF(t) (t)
#endif
#define ZKW_PACK_MAX_WAVES (ZKW_PACK_THREADS / 64u)

static __device__ __forceinline__ void pack_copy(uint4* dst, const uint4* src, u64 n, u32 t, u32 nt) {
  u64 i = t;
  for (; i + 3ull * nt < n; i += 4ull * nt) {  // four loads in flight per thread
    const uint4 a = src[i], b = src[i + nt], c = src[i + 2ull * nt], d = src[i + 3ull * nt];
    dst[i] = a; dst[i + nt] = b; dst[i + 2ull * nt] = c; dst[i + 3ull * nt] = d;
  }
  for (; i < n; i += nt) dst[i] = src[i];
}

// exclusive rank of this thread's flag among the flags of the workgroup's threads (thread order) + their total
static __device__ __forceinline__ u32 pack_flag_scan(bool f, u32* wave_counts, u32 t, u32 nt, u32& total) {
  const unsigned long long m = __ballot(f ? 1 : 0);
  const u32 lane = ZKW_PACK_LANE(t), wv = ZKW_PACK_WAVE_OF(t);
  const u32 below = (u32)__popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_counts[wv] = (u32)__popcll(m);
  __syncthreads();
  u32 off = 0, tot = 0;
  const u32 nw = (nt + 63u) / 64u;
#ifdef ZKW_WIDE
  for (u32 i = 0; i < nw; i++) {
    const u32 c = wave_counts[i];
    if (i < wv) off += c;
    tot += c;
  }
#else
  (void)nw;
  tot = wave_counts[wv];
#endif
  __syncthreads();
  total = tot;
  return off + below;
}

// the same for three small counts per thread at once (c[k] <= 4): exclusive ranks off[k] in thread order, totals tot[k].  The wave-level
// prefix sums run on the counts packed into one u32 (10-bit fields: a wave's total is at most 256)
static __device__ __forceinline__ void pack_count_scan3(const u32 c[3], u32 (*wave_tot)[3], u32 t, u32 nt, u32 off[3], u32 tot[3]) {
  const u32 lane = ZKW_PACK_LANE(t), wv = ZKW_PACK_WAVE_OF(t);
  const u32 packed = c[0] | (c[1] << 10) | (c[2] << 20);
  u32 v = packed;
#ifdef ZKW_WIDE
  for (u32 d = 1; d < 64u; d <<= 1) {
    const u32 o = (u32)__shfl((int)v, (int)(lane >= d ? lane - d : lane));
    if (lane >= d) v += o;
  }
  const u32 wt = (u32)__shfl((int)v, 63);
#else
  const u32 wt = v;
#endif
  const u32 excl = v - packed;
  if (lane == 0) { wave_tot[wv][0] = wt & 1023u; wave_tot[wv][1] = (wt >> 10) & 1023u; wave_tot[wv][2] = wt >> 20; }
  __syncthreads();
  const u32 nw = (nt + 63u) / 64u;
  for (int k = 0; k < 3; k++) { off[k] = (excl >> (10 * k)) & 1023u; tot[k] = 0; }
  for (u32 i = 0; i < nw; i++)
    for (int k = 0; k < 3; k++) {
      const u32 x = wave_tot[i][k];
      if (i < wv) off[k] += x;
      tot[k] += x;
    }
  __syncthreads();
}

// sum of one value per thread over the workgroup
static __device__ __forceinline__ u32 pack_block_sum(u32 mine, u32* sums, u32 t, u32 nt) {
  sums[t] = mine;
  __syncthreads();
  u32 n = 0;
  for (u32 i = 0; i < nt; i++) n += sums[i];
  __syncthreads();
  return n;
}

// which of its parts a register delta (bytes 0..15 `lo`, bytes 16..31 `hi`) travels with under ZKW_PACK_SPARSE_DELTAS
static __device__ __forceinline__ bool pack_delta_has1(uint4 lo) { return (lo.z | lo.w) != 0; }
static __device__ __forceinline__ bool pack_delta_has2(uint4 hi) { return (hi.x | hi.y | hi.z | hi.w) != 0; }

__global__ void __launch_bounds__(ZKW_PACK_THREADS) zkw_pack_kernel(zkw_pack_args A) {
  __shared__ u32 s_wave_counts[4][ZKW_PACK_MAX_WAVES + 1];
  __shared__ u32 s_sums[ZKW_PACK_THREADS];
  __shared__ u32 s_bcast[4];
  __shared__ u32 s_wave_tot[ZKW_PACK_MAX_WAVES + 1][3];
  const u32 t = threadIdx.x, nt = blockDim.x;
  const bool one = A.only_wave != 0xffffffffu;
  const u32 total_waves = one ? 1u : A.wave_base[A.n_batches];
  for (u32 gw = blockIdx.x; gw < total_waves; gw += gridDim.x) {
    u32 b = 0, w = A.only_wave;
    if (!one) {  // largest b with wave_base[b] <= gw
      u32 lo = 0, hi = A.n_batches;
      while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (A.wave_base[mid] <= gw) lo = mid; else hi = mid;
      }
      b = lo;
      w = gw - A.wave_base[lo];
    }
    const zkw_kparams ZKW_CONST_AS& P = *(const zkw_kparams ZKW_CONST_AS*)A.kp[b];
    const u32 L = P.L;
    const u32 max_cyc = P.wave_cycles[w] < P.max_cycles ? P.wave_cycles[w] : P.max_cycles;
    const u32* cur = P.cursors + (u64)w * 4;
    const u32 n_mem = cur[0] < P.cap_mem ? cur[0] : P.cap_mem, n_log = cur[1] < P.cap_log ? cur[1] : P.cap_log;
    const u32 n_aux = cur[2] < P.cap_aux ? cur[2] : P.cap_aux, n_delta = cur[3] < P.cap_delta ? cur[3] : P.cap_delta;
    const uint4* mem_hdr = P.mem_stream + (u64)w * P.cap_mem * 3;
    const uint4* aux_src = P.aux_stream + (u64)w * P.cap_aux * 16;
    // ---- count: memory queries that carry a value (zkw_pack_has_value), units of the aux records ----
    u32 my_val = 0, my_aux = 0, my_page = 0;
    for (u32 base = 0; base < n_mem; base += 4u * nt) {
      u32 hw[4];
      for (u32 j = 0; j < 4; j++) {
        const u32 i = base + j * nt + t;
        hw[j] = i < n_mem ? mem_hdr[i].w : (ZKW_MEM_CODE << 16);  // (beyond the end: a Code read of the VM itself — no value, no page)
      }
      for (u32 j = 0; j < 4; j++) {
        const u32 i = base + j * nt + t;
        my_val += i < n_mem && zkw_pack_has_value(hw[j], A.flags) ? 1u : 0u;
        my_page += i < n_mem && zkw_pack_has_page(hw[j], A.flags) ? 1u : 0u;
      }
    }
    for (u32 i = t; i < n_aux; i += nt) my_aux += zkw_aux_used_units(aux_src[(u64)i * 16].x & 0xffu);
    const u32 n_val = pack_block_sum(my_val, s_sums, t, nt);
    const u32 aux_units = pack_block_sum(my_aux, s_sums, t, nt);
    const u32 n_page = pack_block_sum(my_page, s_sums, t, nt);
    const uint4* dsrc_lo = P.deltas + (u64)w * P.cap_delta * 2;
    const uint4* dsrc_hi = dsrc_lo + P.cap_delta;
    const uint4* tsrc = P.tails + (u64)w * P.max_cycles * L;
    const u64 n_t = (u64)max_cyc * L;
    zkw_pack_counts C;
    C.max_cyc = max_cyc; C.L = L; C.n_delta = n_delta; C.n_mem = n_mem; C.n_page = n_page; C.n_val = n_val; C.n_log = n_log; C.aux_units = aux_units;
    C.n_d1 = C.n_d2 = C.n_tx = C.n_ty = C.n_tz = 0;
    // (the counting passes keep four records per thread in flight: the link leaves the kernel time to read its input twice, but only
    // if the reads overlap — a few dozen workgroups, each load a trip to HBM)
    if (A.flags & ZKW_PACK_SPARSE_DELTAS) {  // deltas with bytes 8..15, with bytes 16..31
      u32 m1 = 0, m2 = 0;
      for (u32 base = 0; base < n_delta; base += 4u * nt) {
        uint4 lo[4], hi[4];
        for (u32 j = 0; j < 4; j++) {
          const u32 i = base + j * nt + t;
          lo[j] = i < n_delta ? dsrc_lo[i] : make_uint4(0, 0, 0, 0);
          hi[j] = i < n_delta ? dsrc_hi[i] : make_uint4(0, 0, 0, 0);
        }
        for (u32 j = 0; j < 4; j++) {
          m1 += pack_delta_has1(lo[j]) ? 1u : 0u;
          m2 += pack_delta_has2(hi[j]) ? 1u : 0u;
        }
      }
      C.n_d1 = pack_block_sum(m1, s_sums, t, nt);
      C.n_d2 = pack_block_sum(m2, s_sums, t, nt);
    }
    if (A.flags & ZKW_PACK_DELTA_TAILS) {  // tails whose pointer bitmap / pc, sp / ergs are not what their predecessor predicts
      u32 mx = 0, my = 0, mz = 0;
      for (u64 base = 0; base < n_t; base += 4ull * nt) {
        uint4 e[4], prev[4];
        bool in[4];
        for (u32 j = 0; j < 4; j++) {
          const u64 i = base + (u64)j * nt + t;
          in[j] = i < n_t;
          e[j] = in[j] ? tsrc[i] : make_uint4(0, 0, 0, 0);
          prev[j] = in[j] && i >= L ? tsrc[i - L] : make_uint4(0, 0, 0, 0);
        }
        for (u32 j = 0; j < 4; j++) {
          const u64 i = base + (u64)j * nt + t;
          const u32 tw = in[j] ? zkw_pack_tail_word(e[j], prev[j], i < L) : (ZKW_TW_X_SAME | ZKW_TW_Y_PRED);
          mx += zkw_tw_has_x(tw) ? 1u : 0u; my += zkw_tw_has_y(tw) ? 1u : 0u; mz += zkw_tw_has_z(tw) ? 1u : 0u;
        }
      }
      C.n_tx = pack_block_sum(mx, s_sums, t, nt);
      C.n_ty = pack_block_sum(my, s_sums, t, nt);
      C.n_tz = pack_block_sum(mz, s_sums, t, nt);
    }
    // ---- allocate the wave's extent ----
    const u64 units64 = zkw_pack_wave_units(&C, A.flags);
    if (t == 0) {
      u32 off = 0;
      if (units64 < 0xffffffffull) {
        off = atomicAdd(A.state, (u32)units64);
        if ((u64)off + units64 > A.dst_units) off = 0;
      }
      if (off == 0) atomicOr(A.state + 1, 1u);
      s_bcast[0] = off;
    }
    __syncthreads();
    const u32 off = s_bcast[0];
    __syncthreads();
    if (t == 0) {  // the wave's table entry
      uint4* e = A.dst + A.wave_table + (u64)(one ? 0u : gw) * ZKW_PACK_WAVE_UNITS;
      e[0] = make_uint4(off, max_cyc, n_delta, n_mem);
      e[1] = make_uint4(n_val, n_log, n_aux, aux_units);
      e[2] = make_uint4((u32)units64, n_page, C.n_d1, C.n_d2);
      e[3] = make_uint4(C.n_tx, C.n_ty, C.n_tz, 0);
    }
    // ---- the per-instance sections of this wave's instances ----
    if (A.with_instances) {
      const zkw_pack_batch pb = A.batches[b];
      const u32 i0 = w * L, i1 = (i0 + L < P.n_instances) ? i0 + L : P.n_instances;
      pack_copy(A.dst + pb.scalars_off + (u64)i0 * 8, (const uint4*)P.scalars + (u64)i0 * 8, (u64)(i1 - i0) * 8, t, nt);
      pack_copy(A.dst + pb.regs_off + (u64)w * ZKW_REG_CHUNKS * L, P.regs + (u64)w * ZKW_REG_CHUNKS * L, (u64)ZKW_REG_CHUNKS * L, t, nt);
      for (u32 j = t; j < (i1 - i0) * 8u; j += nt) {
        const u32 i = i0 + j / 8u, u = j % 8u;
        u32 depth = P.scalars[i].depth;
        if (depth > P.D) depth = P.D;
        A.dst[pb.entries_off + (u64)i * 8 + u] = ((const uint4*)(P.callstack + (u64)i * (P.D + 1) + depth))[u];
      }
    }
    if (off == 0) continue;  // the block is full: the host sees off == 0 and the overflow flag
    uint4* d = A.dst + off;
    // ---- directory, tails, register deltas: plain extents ----
    pack_copy(d, (const uint4*)(P.dir + ((u64)w * (P.max_cycles + 1)) * 4), (u64)max_cyc + 1, t, nt);
    d += max_cyc + 1;
    {
      if (A.flags & ZKW_PACK_DELTA_TAILS) {  // one u32 per tail (four consecutive tails per thread) + the three lists
        const u64 t4 = zkw_ceil4_64(n_t);
        uint4* pa = d;
        u32* lst[3];
        lst[0] = (u32*)(d + t4);
        lst[1] = (u32*)(d + t4 + zkw_ceil4_64(C.n_tx));
        lst[2] = (u32*)(d + t4 + zkw_ceil4_64(C.n_tx) + zkw_ceil4_64(C.n_ty));
        u32 lbase[3] = {0, 0, 0};
        for (u64 base = 0; base < t4; base += nt) {
          const u64 g = base + t;
          u32 tw[4], c[3] = {0, 0, 0};
          uint4 e[4];
          for (u32 j = 0; j < 4; j++) {
            const u64 i = 4ull * g + j;
            const bool in = g < t4 && i < n_t;
            e[j] = in ? tsrc[i] : make_uint4(0, 0, 0, 0);
            const uint4 prev = in && i >= L ? tsrc[i - L] : make_uint4(0, 0, 0, 0);
            tw[j] = in ? zkw_pack_tail_word(e[j], prev, i < L) : (ZKW_TW_X_SAME | ZKW_TW_Y_PRED);
            c[0] += zkw_tw_has_x(tw[j]) ? 1u : 0u; c[1] += zkw_tw_has_y(tw[j]) ? 1u : 0u; c[2] += zkw_tw_has_z(tw[j]) ? 1u : 0u;
          }
          if (g < t4) pa[g] = make_uint4(tw[0], tw[1], tw[2], tw[3]);
          u32 off3[3], tot3[3];
          pack_count_scan3(c, s_wave_tot, t, nt, off3, tot3);
          for (u32 j = 0; j < 4; j++) {
            if (zkw_tw_has_x(tw[j])) lst[0][lbase[0] + off3[0]++] = e[j].x & 0x00ffffffu;
            if (zkw_tw_has_y(tw[j])) lst[1][lbase[1] + off3[1]++] = e[j].y;
            if (zkw_tw_has_z(tw[j])) lst[2][lbase[2] + off3[2]++] = e[j].z;
          }
          for (int k = 0; k < 3; k++) lbase[k] += tot3[k];
        }
      } else if (A.flags & ZKW_PACK_SLIM_TAILS) {  // x | y | z as u32 planes, the top byte of w (delta mask, high half) as a byte plane; the counts stay behind
        const u64 t4 = (n_t + 3ull) >> 2, t16 = (n_t + 15ull) >> 4;
        uint4 *px = d, *py = d + t4, *pz = d + 2ull * t4, *pb = d + 3ull * t4;
        for (u64 g = t; g < t16; g += nt) {  // sixteen consecutive tails per thread: one 16-byte store of the byte plane
          u32 bytes[4];
          for (u32 q = 0; q < 4; q++) {
            uint4 e[4];
            for (u32 j = 0; j < 4; j++) {
              const u64 i = 16ull * g + 4u * q + j;
              e[j] = i < n_t ? tsrc[i] : make_uint4(0, 0, 0, 0);
            }
            const u64 u = 4ull * g + q;
            if (u < t4) {
              px[u] = make_uint4(e[0].x, e[1].x, e[2].x, e[3].x);
              py[u] = make_uint4(e[0].y, e[1].y, e[2].y, e[3].y);
              pz[u] = make_uint4(e[0].z, e[1].z, e[2].z, e[3].z);
            }
            bytes[q] = (e[0].w >> 24) | ((e[1].w >> 24) << 8) | ((e[2].w >> 24) << 16) | ((e[3].w >> 24) << 24);
          }
          pb[g] = make_uint4(bytes[0], bytes[1], bytes[2], bytes[3]);
        }
      } else {
        pack_copy(d, tsrc, n_t, t, nt);
      }
      d += zkw_pack_tail_units(n_t, &C, A.flags);
    }
    if (A.flags & ZKW_PACK_SPARSE_DELTAS) {  // (layout: zkw_pack_delta_units)
      const u32 nblk = (n_delta + 255u) >> 8;
      uint4* bits = d;
      uint4* p0 = d + 4ull * nblk;                                  // u64 [n_delta]
      unsigned long long* p1 = (unsigned long long*)(p0 + zkw_ceil2_64(n_delta));  // u64 [n_d1]
      uint4* p2 = p0 + zkw_ceil2_64(n_delta) + zkw_ceil2_64(C.n_d1);  // 16 B [n_d2]
      u32 base1 = 0, base2 = 0;
      const u32 g4 = zkw_ceil4(n_delta);
#ifdef ZKW_WIDE
      for (u32 base = 0; base < g4; base += nt) {  // a thread takes four consecutive deltas; a wave one block of the bit planes
        const u32 g = base + t;
        uint4 lo[4], hi[4];
        bool h1[4], h2[4];
        u32 c[3] = {0, 0, 0};
        for (u32 j = 0; j < 4; j++) {
          const u32 i = 4u * g + j;
          const bool in = g < g4 && i < n_delta;
          lo[j] = in ? dsrc_lo[i] : make_uint4(0, 0, 0, 0);
          hi[j] = in ? dsrc_hi[i] : make_uint4(0, 0, 0, 0);
          h1[j] = pack_delta_has1(lo[j]); h2[j] = pack_delta_has2(hi[j]);
          c[0] += h1[j] ? 1u : 0u; c[1] += h2[j] ? 1u : 0u;
        }
        unsigned long long m[8];
        for (u32 j = 0; j < 4; j++) { m[j] = __ballot(h1[j] ? 1 : 0); m[4 + j] = __ballot(h2[j] ? 1 : 0); }
        const u32 lane = ZKW_PACK_LANE(t), blk = g >> 6;
        if (lane < 4u && blk < nblk) bits[4ull * blk + lane] = make_uint4((u32)m[2 * lane], (u32)(m[2 * lane] >> 32), (u32)m[2 * lane + 1], (u32)(m[2 * lane + 1] >> 32));
        if (g < g4) {
          const u64 u = 2ull * g;  // units of the plane of bytes 0..7
          if (u < zkw_ceil2_64(n_delta)) p0[u] = make_uint4(lo[0].x, lo[0].y, lo[1].x, lo[1].y);
          if (u + 1 < zkw_ceil2_64(n_delta)) p0[u + 1] = make_uint4(lo[2].x, lo[2].y, lo[3].x, lo[3].y);
        }
        u32 off3[3], tot3[3];
        pack_count_scan3(c, s_wave_tot, t, nt, off3, tot3);
        for (u32 j = 0; j < 4; j++) {
          if (h1[j]) p1[base1 + off3[0]++] = (unsigned long long)lo[j].z | ((unsigned long long)lo[j].w << 32);
          if (h2[j]) p2[base2 + off3[1]++] = hi[j];
        }
        base1 += tot3[0]; base2 += tot3[1];
      }
#else  // (one thread: the same layout, delta by delta)
      (void)g4;
      for (u32 blk = 0; blk < nblk; blk++) {
        unsigned long long m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (u32 r = 0; r < 256u; r++) {
          const u32 i = 256u * blk + r;
          if (i >= n_delta) break;
          const uint4 lo = dsrc_lo[i], hi = dsrc_hi[i];
          ((unsigned long long*)p0)[i] = (unsigned long long)lo.x | ((unsigned long long)lo.y << 32);
          if (pack_delta_has1(lo)) { m[r & 3u] |= 1ull << (r >> 2); p1[base1++] = (unsigned long long)lo.z | ((unsigned long long)lo.w << 32); }
          if (pack_delta_has2(hi)) { m[4u + (r & 3u)] |= 1ull << (r >> 2); p2[base2++] = hi; }
        }
        for (u32 u = 0; u < 4u; u++) bits[4ull * blk + u] = make_uint4((u32)m[2 * u], (u32)(m[2 * u] >> 32), (u32)m[2 * u + 1], (u32)(m[2 * u + 1] >> 32));
      }
      if (n_delta & 1u) ((unsigned long long*)p0)[n_delta] = 0;
#endif
      d += zkw_pack_delta_units(&C, A.flags);
    } else {
      pack_copy(d, dsrc_lo, n_delta, t, nt);
      pack_copy(d + n_delta, dsrc_hi, n_delta, t, nt);
      d += 2ull * n_delta;
    }
    // ---- memory queries: 12-byte headers as three u32 planes, values of the non-code queries only ----
    {
      const u32 q4 = zkw_ceil4(n_mem), p4 = zkw_ceil4(n_page);
      u32* pl_page = (u32*)d;  // the page list: one u32 per query that carries its page
      uint4* pl_index = d + p4;
      uint4* pl_misc = d + p4 + q4;
      uint4* v_lo = d + p4 + 2ull * q4;
      uint4* v_hi = v_lo + n_val;
      const uint4* src_lo = mem_hdr + P.cap_mem;
      const uint4* src_hi = mem_hdr + 2ull * P.cap_mem;
      u32 vbase = 0, pbase = 0;
      for (u32 base = 0; base < q4; base += nt) {  // a thread takes four consecutive records: one 16-byte store per plane
        const u32 g = base + t;
        uint4 h[4];
        bool has[4], hasp[4];
        for (int j = 0; j < 4; j++) {
          const u32 i = 4u * g + (u32)j;
          const bool in = g < q4 && i < n_mem;
          h[j] = in ? mem_hdr[i] : make_uint4(0, 0, 0, 0);
          has[j] = in && zkw_pack_has_value(h[j].w, A.flags);
          hasp[j] = in && zkw_pack_has_page(h[j].w, A.flags);
        }
        if (g < q4) {
          pl_index[g] = make_uint4(h[0].z, h[1].z, h[2].z, h[3].z);
          pl_misc[g] = make_uint4((h[0].w & 0x00ffffffu) | (h[0].x << 24), (h[1].w & 0x00ffffffu) | (h[1].x << 24), (h[2].w & 0x00ffffffu) | (h[2].x << 24),
                                  (h[3].w & 0x00ffffffu) | (h[3].x << 24));
        }
        // value positions: records in stream order = thread order, then j
        u32 tot[4], rk[4], tile = 0, mine_before = 0;
        for (int j = 0; j < 4; j++) rk[j] = pack_flag_scan(has[j], s_wave_counts[j], t, nt, tot[j]);
        // rank of record (t, j) = sum over j' of flags of threads below t + flags (t, j' < j)
        u32 below_threads = rk[0] + rk[1] + rk[2] + rk[3];
        for (int j = 0; j < 4; j++) {
          if (has[j]) {
            const u32 at = vbase + below_threads + mine_before;
            const u32 i = 4u * g + (u32)j;
            v_lo[at] = src_lo[i];
            v_hi[at] = src_hi[i];
            mine_before++;
          }
          tile += tot[j];
        }
        vbase += tile;
        // the page list, the same way (records in stream order = thread order, then j)
        tile = 0; mine_before = 0;
        for (int j = 0; j < 4; j++) rk[j] = pack_flag_scan(hasp[j], s_wave_counts[j], t, nt, tot[j]);
        below_threads = rk[0] + rk[1] + rk[2] + rk[3];
        for (int j = 0; j < 4; j++) {
          if (hasp[j]) {
            pl_page[pbase + below_threads + mine_before] = h[j].y;
            mine_before++;
          }
          tile += tot[j];
        }
        pbase += tile;
      }
      d = v_hi + n_val;
    }
    // ---- log queries ----
    pack_copy(d, P.log_stream + (u64)w * P.cap_log * 8, (u64)n_log * 8, t, nt);
    d += (u64)n_log * 8;
    // ---- aux events: each as long as its type uses ----
    {
      u32 abase = 0;
      for (u32 base = 0; base < n_aux; base += nt) {
        const u32 i = base + t;
        const u32 used = i < n_aux ? zkw_aux_used_units(aux_src[(u64)i * 16].x & 0xffu) : 0u;
        s_sums[t] = used;
        __syncthreads();
        u32 before = 0, tile = 0;
        for (u32 k = 0; k < nt; k++) {
          const u32 c = s_sums[k];
          if (k < t) before += c;
          tile += c;
        }
        __syncthreads();
        for (u32 u = 0; u < used; u++) d[abase + before + u] = aux_src[(u64)i * 16 + u];
        abase += tile;
      }
    }
  }
}

extern "C" hipError_t zkw_launch_pack(const zkw_pack_args* A, uint32_t wave_threads, uint32_t blocks, hipStream_t stream) {
  const uint32_t waves = A->only_wave != 0xffffffffu ? 1u : A->wave_base[A->n_batches];
  if (waves == 0) return hipSuccess;
  uint32_t g = blocks ? blocks : 64u;
  if (g > waves) g = waves;
  hipLaunchKernelGGL(zkw_pack_kernel, dim3(g), dim3(wave_threads > 1 ? ZKW_PACK_THREADS : 1), 0, stream, *A);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// zkw_restage_kernel (zkw_pack.h): VmLocalStates + instance-major heap images -> the pristine device images of a batch
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) zkw_restage_kernel(zkw_restage_params R) {
  const u64 stride = (u64)gridDim.x * blockDim.x, t0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  // states: one thread per instance (lane = instance: the 30 register chunks of adjacent threads are adjacent)
  for (u64 i = t0; i < R.n_instances; i += stride) {
    const zkw_vm_local_state* st = R.states + i;
    const u32 w = (u32)(i / R.L), l = (u32)(i % R.L);
    const uint4* rv = (const uint4*)st->registers;
    for (u32 ch = 0; ch < ZKW_REG_CHUNKS; ch++) R.regs0[((u64)w * ZKW_REG_CHUNKS + ch) * R.L + l] = rv[ch];
    // the scalar half (the same mapping as format_instance in zkw_runtime.cpp; the arena bookkeeping — first dynamic page,
    // initial slots — follows the state, the rest of the row is what the upload put there)
    zkw_dev_scalars sc = R.scalars0[i];
    const u32* pcw = (const u32*)st->previous_code_word.l;
    for (int k = 0; k < 8; k++) sc.prev_code_word[k] = pcw[k];
    const u32* cx = (const u32*)st->context_u128_register;
    for (int k = 0; k < 4; k++) sc.ctx_u128_reg[k] = cx[k];
    sc.ptr_bitmap = st->register_ptr_bitmap & 0x7fffu;
    sc.flags = (st->flags & 7u) | (st->pending_exception ? 8u : 0u);
    sc.prev_code_page = st->previous_code_memory_page;
    sc.timestamp = st->timestamp;
    sc.cycle_counter = st->monotonic_cycle_counter;
    sc.spent_pubdata = st->spent_pubdata_counter;
    sc.memory_page_counter = st->memory_page_counter;
    sc.absolute_execution_step = st->absolute_execution_step;
    sc.ergs_per_pubdata = st->current_ergs_per_pubdata_byte;
    sc.tx_number = st->tx_number_in_block;
    sc.prev_super_pc = st->previous_super_pc;
    sc.depth = st->callstack_depth;
    sc.status = ZKW_STATUS_RUNNING;
    sc.n_cycles = 0;
    sc.first_dynamic_page = st->memory_page_counter;
    R.scalars0[i] = sc;
    // callstack.current: the ABI struct of the row at `depth`; blob, arena slot and journal mark of the row stay (the host has
    // checked that code page and base page are the uploaded ones)
    zkw_dev_entry* row = R.callstack0 + i * (R.D + 1) + st->callstack_depth;
    zkw_callstack_entry e = st->current;
    e.reserved0 = 0;
    e.reserved1 = 0;
    row->e = e;
    // a restaged heap image is image_words long for every instance, whatever the instance had uploaded: the page's mark
    // (words at and beyond it read as zero, heap_read_at) follows the new image
    if (R.heaps && R.image_words) R.frames0[i * R.F].heap_hwm = R.image_words;
  }
  // heap images: out[((w * words + k) * 2 + half) * L + l] = in[(i * words + k) * 2 + half]; consecutive threads = consecutive lanes
  // of one (word, half): the stores are whole lines, the loads 16 bytes out of each lane's own image
  if (R.heaps && R.image_words) {
    const u64 per_wave = (u64)R.image_words * 2u * R.L, total = per_wave * R.n_waves;
    for (u64 o = t0; o < total; o += stride) {
      const u32 w = (u32)(o / per_wave);
      const u64 r = o % per_wave;
      const u32 l = (u32)(r % R.L);
      const u64 kh = r / R.L;  // word * 2 + half
      const u64 i = (u64)w * R.L + l;
      R.heap0[o] = i < R.n_instances ? R.heaps[i * R.image_words * 2u + kh] : make_uint4(0, 0, 0, 0);
    }
  }
}

extern "C" hipError_t zkw_launch_restage(const zkw_restage_params* R, uint32_t wave_threads, hipStream_t stream) {
  const uint64_t work = (uint64_t)R->n_waves * R->image_words * 2u * R->L + R->n_instances;
  uint32_t blocks = (uint32_t)((work + 4u * 256u - 1) / (4u * 256u));
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zkw_restage_kernel, dim3(wave_threads > 1 ? blocks : 1), dim3(wave_threads > 1 ? 256 : 1), 0, stream, *R);
  return hipGetLastError();
}
