// Queue commitments: memory / log / decommit queue digests per VM instance.
//
// The reference crate has NO sponge or queue commitment (only a comment, far_call.rs:29-32, and a
// dead stub, vm_state/aux_data.rs:1-6): downstream circuits own that.  This file implements the
// build's OWN spec ("ZKW-GL-sponge v2") so that the north star's
// "algebraic sponge absorb over the emitted queues" and the multi-GPU final reduction have something
// concrete and testable; oracle/commit.hpp restates the same spec on the CPU.  PARITY UNPINNED w.r.t.
// any real zkEVM circuit format.
//
//   field   Goldilocks p = 2^64 - 2^32 + 1, canonical representatives
//   P       Poseidon2-shaped permutation, t = 12, x^7, 4 + 22 + 4 rounds, external layer
//           circ(2 M4, M4, M4), internal layer J + diag(2^i), constants from splitmix64("zkwGLv1")
//   memory / log record   its u32 words packed into elements below 2^56 (zkw_goldilocks.hip.h: gl_chain_record); every
//           block of 7 elements is one permutation  tail' = P(e || tail || (j+1) | queue << 40 | block << 48)[0..4]
//   decommit record       leaf = sponge (rate 8, domain (type, length)) over code hash | blob length | blob digest, cached
//           per preimage at upload;  tail' = P(leaf || tail || j+1 || queue || timestamp | fresh << 32 || page)[0..4]
//   code words (blob digests, upload only)  sponge leaf per word, chunk chains, one chain over the chunk tails
//
// Parallel structure on the GPU: a bucket pass turns the lane tags of each wave stream into per-instance index lists,
// and the sequential chains run one instance per lane in lockstep, all queues of a step in one launch.
#include <hip/hip_runtime.h>

#include "zkw_device.h"
#include "zkw_commit.h"

typedef uint32_t u32;
typedef uint64_t u64;
#define ZD __device__ __forceinline__

#include "zkw_goldilocks.hip.h"

// Register bound of the permutation kernels.  The field multiplication (gl_mulred) names v112..v118 in its asm, so these
// kernels are allocated 128 registers = 4 waves per SIMD whatever the bound says; a tighter bound (it was 6) only made
// the compiler squeeze everything else into 80 registers and spill 82 of them (332 B of scratch per lane).
#ifndef ZKW_CHAIN_MIN_WAVES
#define ZKW_CHAIN_MIN_WAVES 4
#endif

// orders one wavefront's LDS stores before its later cross-lane LDS reads
ZD void zkw_commit_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------

// one stream record per thread -> leaf[wave][pos] (4 x u64)
// Leaves exist only for the code words of the blobs (Q = ZKW_QUEUE_CODE_WORDS, at upload): the records of the memory and
// log queues are the inputs of their chain permutations, the decommit leaves are cached per preimage.
template <int Q>
__global__ void __attribute__((amdgpu_waves_per_eu(ZKW_CHAIN_MIN_WAVES, 8))) zkw_leaf_kernel(zkw_fused_table T) {
  const zkw_commit_params ZKW_CONST_AS& C = *(const zkw_commit_params ZKW_CONST_AS*)T.p[blockIdx.z];
  if (blockIdx.y >= C.n_waves) return;
  const u32 wave = blockIdx.y;
  u32 n = C.n_override;
  if (!n) n = C.cursors[wave * 4 + C.queue] < C.cap ? C.cursors[wave * 4 + C.queue] : C.cap;
  // (the decommit queue has no leaf pass: its leaves are cached per preimage, zkw_midstate_kernel)
  for (u32 pos = blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += gridDim.x * blockDim.x) {
    u64 out[4];
    {  // code words of the blobs: stream = blob words (2 x uint4 per word), wave = 0
      const uint4 lo = C.stream[(u64)pos * 2], hi = C.stream[(u64)pos * 2 + 1];
      u64 f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      gl_leaf<8>(C.rc, ZKW_LEAF_CODE_WORD, f, out);
    }
    u64* dst = C.leaves + ((u64)wave * C.cap + pos) * 4;
    dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
  }
}

// per wave: stable partition of the wave's stream by lane tag -> idx[inst][j], count[inst].
// One tile of blockDim.x records per step: six ballots give every thread the mask of the threads
// holding a record of the same lane, its rank inside that group is a popcount, and the group's
// leader bumps the per-lane running count in LDS.  Records of cycles the instance did not complete
// (failed cycles) are dropped through the stream directory: record p of lane l counts iff
// p < dir[n_cycles[l]].
// bucket / chain launches cover several queues of a batch at once: T.reserved[1] = mask of the queues, blockIdx.z = which
// of them; T.p[batch] then points to the batch's parameter block of queue 0.  (mask 0: T.p[batch] is the block itself.)
ZD const zkw_commit_params ZKW_CONST_AS& zkw_commit_block(const zkw_fused_table& T, u32 batch, u32 z) {
  const zkw_commit_params ZKW_CONST_AS* p = (const zkw_commit_params ZKW_CONST_AS*)T.p[batch];
  u32 mask = T.reserved[1];
  if (!mask) return *p;
  for (u32 k = 0; k < z; k++) mask &= mask - 1u;
  return p[(u32)__ffsll((long long)mask) - 1u];
}

__global__ void zkw_bucket_kernel(zkw_fused_table T) {
  const zkw_commit_params ZKW_CONST_AS& C = zkw_commit_block(T, blockIdx.y, blockIdx.z);
  __shared__ u32 s_count[ZKW_WAVE];
  __shared__ u32 s_limit[ZKW_WAVE];
  __shared__ u32 s_off[ZKW_WAVE];
  const u32 wave = blockIdx.x;
  const u32 tid = threadIdx.x;
  if (wave >= C.n_waves) return;  // uniform per workgroup
  const u32 rec_bytes = C.queue == ZKW_QUEUE_MEMORY ? 48u : (C.queue == ZKW_QUEUE_LOG ? 128u : 256u);
  const u32 type_mask = C.aux_type_mask ? C.aux_type_mask : (1u << ZKW_AUX_DECOMMIT);
  const uint8_t* base = (const uint8_t*)C.stream + (u64)wave * C.cap * rec_bytes;
  const u32 n = C.cursors[wave * 4 + C.queue] < C.cap ? C.cursors[wave * 4 + C.queue] : C.cap;
  const uint32_t* dir = C.dir + (u64)wave * (C.max_cycles + 1) * 4;
  for (u32 l = tid; l < ZKW_WAVE; l += blockDim.x) {
    const u32 inst = wave * C.L + l;
    u32 lim = 0;
    if (l < C.L && inst < C.n_instances) lim = dir[(u64)C.scalars[inst].n_cycles * 4 + C.queue];
    s_limit[l] = lim < n ? lim : n;
    s_count[l] = 0;
  }
  __syncthreads();
  // pooled lists: a counting pass first (the same walk, nothing placed), then every lane's list starts where the previous
  // lane's ends inside the wave's `cap` entries
  for (u32 pass = C.pooled ? 0u : 1u; pass < 2u; pass++) {
    for (u32 tile = 0; tile < n; tile += blockDim.x) {
      const u32 p = tile + tid;
      u32 tag = 0;
      bool keep = false;
      if (p < n) {
        u32 type = ZKW_AUX_DECOMMIT;
        if (C.queue == ZKW_QUEUE_MEMORY) {
          tag = base[(u64)p * 16 + 12];  // plane 0 (headers) of the wave's memory stream
        } else if (C.queue == ZKW_QUEUE_LOG) {
          tag = base[(u64)p * 128 + 126];
        } else {
          tag = base[(u64)p * 256 + 1];
          type = base[(u64)p * 256];
        }
        keep = tag < ZKW_WAVE && p < s_limit[tag & (ZKW_WAVE - 1)] && ((type_mask >> (type & 31u)) & 1u);
      }
      tag &= ZKW_WAVE - 1;
      u64 same = __ballot(keep);
#pragma unroll
      for (int bit = 0; bit < 6; bit++) {
        const bool b = (tag >> bit) & 1u;
        const u64 m = __ballot(keep && b);
        same &= b ? m : ~m;
      }
      ZKW_DIV_IF(keep) {
        const u32 rank = (u32)__popcll(same & ((1ull << (tid & 63u)) - 1ull));
        const u32 before = s_count[tag];
        const u32 inst = wave * C.L + tag;
        ZKW_LOCKSTEP();  // (every lane of a group has read the running count before the group's leader advances it)
        if (pass == 1u) {
          if (C.pooled) C.idx[(u64)wave * C.cap + s_off[tag] + before + rank] = p;
          else if (before + rank < C.per_instance_cap) C.idx[(u64)inst * C.per_instance_cap + before + rank] = p;
        }
        if (rank == 0) s_count[tag] = before + (u32)__popcll(same);
      }
      __syncthreads();
    }
    if (pass == 0u) {
      if (tid == 0) {
        u32 run = 0;
        for (u32 l = 0; l < ZKW_WAVE; l++) {
          s_off[l] = run;
          run += s_count[l];
          s_count[l] = 0;
        }
      }
      __syncthreads();
    }
  }
  for (u32 l = tid; l < C.L; l += blockDim.x) {
    const u32 inst = wave * C.L + l;
    if (inst >= C.n_instances) continue;
    if (C.pooled) {
      C.counts[inst] = s_count[l];
      C.offs[inst] = wave * C.cap + s_off[l];
    } else {
      C.counts[inst] = s_count[l] < C.per_instance_cap ? s_count[l] : C.per_instance_cap;
    }
  }
}

// one instance per lane: sequential chain over its leaves
__global__ void __attribute__((amdgpu_waves_per_eu(ZKW_CHAIN_MIN_WAVES, 8))) zkw_chain_kernel(zkw_fused_table T) {
  const zkw_commit_params ZKW_CONST_AS& C = zkw_commit_block(T, blockIdx.y, blockIdx.z);
  const u32 wave = blockIdx.x;
  const u32 lane = threadIdx.x;
  const u32 inst = wave * C.L + lane;
  if (wave >= C.n_waves || lane >= C.L || inst >= C.n_instances) return;
  const u32 cnt = C.counts[inst];
  const u32* idx = C.pooled ? C.idx + C.offs[inst] : C.idx + (u64)inst * C.per_instance_cap;
  u64 tail[4] = {0, 0, 0, 0};
  if (C.queue == ZKW_QUEUE_DECOMMIT) {
    // one permutation per decommit: the leaf depends only on the code (hash, length, blob digest) and was computed per
    // preimage at upload (zkw_midstate_kernel); the per-record fields ride in the two spare elements of the chain step
    for (u32 j = 0; j < cnt; j++) {
      const uint4* e = C.stream + ((u64)wave * C.cap + idx[j]) * 16;
      const uint4 h = e[0];
      const u32 pre = e[3].x;
      const u64* ms = C.midstates + (u64)pre * 12;
      const u64 leaf[4] = {ms[0], ms[1], ms[2], ms[3]};
      gl_chain_step(C.rc, leaf, tail, (u64)j + 1, C.queue, (u64)h.y | ((u64)((h.x >> 24) & 0xffu) << 32), (u64)h.z);
    }
  } else if (C.queue == ZKW_QUEUE_MEMORY) {
    // the record is the input of the chain permutation (zkw_goldilocks.hip.h: gl_chain_record): one permutation per query
    for (u32 j = 0; j < cnt; j++) {
      const uint4* e = C.stream + (u64)wave * C.cap * 3 + idx[j];  // three planes per wave: header | value low | value high
      const uint4 h = e[0], lo = e[C.cap], hi = e[2 * (u64)C.cap];
      const u32 w[12] = {h.x, h.y, h.z, (h.w >> 16) & 0xffu, lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      gl_chain_record<6>(C.rc, w, tail, (u64)j + 1, ZKW_QUEUE_MEMORY);
    }
  } else {  // log queue: 32 words -> 19 elements -> three permutations per query
    for (u32 j = 0; j < cnt; j++) {
      const uint4* e = C.stream + ((u64)wave * C.cap + idx[j]) * 8;
      u32 w[32];
      const uint4 a6 = e[6], a7 = e[7];
      w[0] = a7.y;                                              // timestamp
      w[1] = a7.z & 0xffffu;                                    // tx number
      w[2] = ((a7.z >> 16) & 0xffu) | (((a7.z >> 24) & 0xffu) << 8) | ((a7.w & 0xffu) << 16) | (((a7.w >> 8) & 0xffu) << 24);  // aux|shard|bools|kind
      w[3] = a6.x; w[4] = a6.y; w[5] = a6.z; w[6] = a6.w; w[7] = a7.x;  // address
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const uint4 v = e[q];
        w[8 + 4 * q] = v.x; w[9 + 4 * q] = v.y; w[10 + 4 * q] = v.z; w[11 + 4 * q] = v.w;
      }
      gl_chain_record<16>(C.rc, w, tail, (u64)j + 1, ZKW_QUEUE_LOG);
    }
  }
  u64* dst = C.out + ((u64)inst * ZKW_QUEUE_COUNT + C.queue) * 4;
  dst[0] = tail[0]; dst[1] = tail[1]; dst[2] = tail[2]; dst[3] = tail[3];
}

// blob digests, two levels (run once per upload and new bytecode): a sequential chain over 64-word chunks in parallel
// (one chunk per thread), then one chain per blob over its chunk tails — a 2000-word blob is 64 + 32 sequential
// permutations instead of 2000.
__global__ void zkw_blob_chunk_kernel(zkw_fused_table T) {
  const zkw_commit_params ZKW_CONST_AS& C = *(const zkw_commit_params ZKW_CONST_AS*)T.p[0];
  const u32 b = blockIdx.y;
  if (b >= C.n_blobs) return;
  const uint2 d = C.blob_dir[b];
  const u32 n_chunks = (d.y + ZKW_BLOB_CHUNK_WORDS - 1) / ZKW_BLOB_CHUNK_WORDS;
  for (u32 c = blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += gridDim.x * blockDim.x) {
    const u32 first = c * ZKW_BLOB_CHUNK_WORDS;
    const u32 cnt = d.y - first < ZKW_BLOB_CHUNK_WORDS ? d.y - first : ZKW_BLOB_CHUNK_WORDS;
    u64 tail[4] = {0, 0, 0, 0};
    for (u32 j = 0; j < cnt; j++) {
      const u64* lf = C.leaves + ((u64)d.x + first + j) * 4;
      const u64 leaf[4] = {lf[0], lf[1], lf[2], lf[3]};
      gl_chain_step(C.rc, leaf, tail, (u64)j + 1, ZKW_QUEUE_ID_BLOB);
    }
    u64* dst = C.chunk_tails + ((u64)(d.x / ZKW_BLOB_CHUNK_WORDS) + b + c) * 4;
    dst[0] = tail[0]; dst[1] = tail[1]; dst[2] = tail[2]; dst[3] = tail[3];
  }
}
__global__ void zkw_blob_chain_kernel(zkw_fused_table T) {
  const zkw_commit_params ZKW_CONST_AS& C = *(const zkw_commit_params ZKW_CONST_AS*)T.p[0];
  for (u32 b = blockIdx.x * blockDim.x + threadIdx.x; b < C.n_blobs; b += gridDim.x * blockDim.x) {
    const uint2 d = C.blob_dir[b];
    const u32 n_chunks = (d.y + ZKW_BLOB_CHUNK_WORDS - 1) / ZKW_BLOB_CHUNK_WORDS;
    u64 tail[4] = {0, 0, 0, 0};
    for (u32 c = 0; c < n_chunks; c++) {
      const u64* lf = C.chunk_tails + ((u64)(d.x / ZKW_BLOB_CHUNK_WORDS) + b + c) * 4;
      const u64 leaf[4] = {lf[0], lf[1], lf[2], lf[3]};
      gl_chain_step(C.rc, leaf, tail, (u64)c + 1, ZKW_QUEUE_ID_BLOB_TOP);
    }
    u64* dst = C.out + (u64)b * 4;
    dst[0] = tail[0]; dst[1] = tail[1]; dst[2] = tail[2]; dst[3] = tail[3];
  }
}

// leaf of a decommit record = sponge over what identifies the code: the 8 limbs of the code hash, the length of the blob
// it maps to and the blob's digest (13 elements, two permutations) — once per (hash -> blob) pair and upload; a
// decommit then costs one permutation (chain step).  Stored in the first 4 of the 12 elements of the preimage's slot.
__global__ void zkw_midstate_kernel(zkw_fused_table T) {
  const zkw_commit_params ZKW_CONST_AS& C = *(const zkw_commit_params ZKW_CONST_AS*)T.p[0];
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < C.n_preimages; i += gridDim.x * blockDim.x) {
    const u32 blob = C.preimages[i].blob;
    const u64* bd = C.blob_digests + (u64)blob * 4;
    u64 f[13];
#pragma unroll
    for (int k = 0; k < 8; k++) f[k] = C.preimages[i].hash[k];
    f[8] = C.blob_dir[blob].y & 0xffffu;
    f[9] = bd[0]; f[10] = bd[1]; f[11] = bd[2]; f[12] = bd[3];
    u64 out[4];
    gl_leaf<13>(C.rc, ZKW_LEAF_DECOMMIT, f, out);
#pragma unroll
    for (int k = 0; k < 12; k++) C.midstates[(u64)i * 12 + k] = k < 4 ? out[k] : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// final net states: one instance per lane.  The walk is the frame discipline of the reference sinks:
//   query            -> history;  writes also push a pending rollback          (storage.rs:100-118, event_sink.rs:140-151)
//   start_frame      -> remember the rollback-stack heights                    (storage.rs:140-143, event_sink.rs:152-155)
//   finish_frame ok  -> the frame's pending rollbacks now belong to its parent (storage.rs:181-185, event_sink.rs:170-174)
//   finish_frame bad -> they are appended to the history in reverse order      (storage.rs:176-180, event_sink.rs:166-169)
// What is left on the event rollback stack at the end is exactly what InMemoryEventSink::flatten keeps
// (event_sink.rs:84-96: insert on forward, remove on rollback, sort by timestamp = emission order).
// ---------------------------------------------------------------------------------------------
__global__ void zkw_netstate_kernel(zkw_fused_table T) {
  const zkw_netstate_params ZKW_CONST_AS& N = *(const zkw_netstate_params ZKW_CONST_AS*)T.p[blockIdx.y];
  const u32 wave = blockIdx.x;
  const u32 lane = threadIdx.x;
  const u32 inst = wave * N.L + lane;
  if (wave >= N.n_waves || lane >= N.L || inst >= N.n_instances) return;
  u32* outc = N.out_counts + (u64)inst * 4;
  const u32 status = N.scalars[inst].status;
  if (status >= ZKW_STATUS_UNKNOWN_CODE_HASH) {
    outc[0] = 0; outc[1] = 0; outc[2] = 0; outc[3] = 2u;  // failed instance: no net state
    return;
  }
  const u32 n_cyc = N.scalars[inst].n_cycles;
  const u32* lidx = N.log_idx + (u64)inst * N.per_log;
  const u32* aidx = N.aux_idx + (u64)inst * N.per_aux;
  const u32 n_l = N.log_cnt[inst], n_a = N.aux_cnt[inst];
  u32* st_hist = N.st_hist + (u64)inst * N.hist_cap;
  u32* ev_hist = N.ev_hist + (u64)inst * N.hist_cap;
  u32* rb_st = N.rb_st + (u64)inst * N.per_log;
  u32* rb_ev = N.rb_ev + (u64)inst * N.per_log;
  u32* marks = N.marks + (u64)inst * N.mark_cap * 2;
  u32 n_sh = 0, n_eh = 0, n_rs = 0, n_re = 0, n_mk = 0, li = 0, ai = 0, flags = 0;
  if (n_l >= N.per_log || n_a >= N.per_aux) flags |= 1u;  // an index list was clipped: the walk would be incomplete
  // frames that were already open when the batch took over (push_bootloader_context / initial callstack depth)
  const u32 depth0 = N.scalars0[inst].depth;
  for (u32 d = 0; d < depth0 && n_mk < N.mark_cap; d++) {
    marks[2 * n_mk] = 0; marks[2 * n_mk + 1] = 0;
    n_mk++;
  }
  const uint4* lbase = N.log_stream + (u64)wave * N.cap_log * 8;
  const uint4* abase = N.aux_stream + (u64)wave * N.cap_aux * 16;
  for (u32 c = 0; c < n_cyc && !flags; c++) {
    const u32 cnt = N.tails[((u64)wave * N.max_cycles + c) * N.L + lane].w;
    u32 nl = (cnt >> 8) & 0xffu, na = (cnt >> 16) & 0xffu;
    while ((nl || na) && !flags) {
      // next record of this cycle by in-cycle sequence number (SURVEY Appendix A order)
      u32 lseq = 0xffffffffu, aseq = 0xffffffffu, lp = 0, ap = 0;
      uint4 l7 = make_uint4(0, 0, 0, 0), a0 = make_uint4(0, 0, 0, 0);
      if (nl && li < n_l) { lp = lidx[li]; l7 = lbase[(u64)lp * 8 + 7]; lseq = l7.w >> 24; }
      if (na && ai < n_a) { ap = aidx[ai]; a0 = abase[(u64)ap * 16]; aseq = (a0.x >> 16) & 0xffu; }
      if (lseq == 0xffffffffu && aseq == 0xffffffffu) { flags |= 1u; break; }
      if (lseq <= aseq) {
        li++; nl--;
        const u32 kind = (l7.w >> 8) & 0xffu;
        if (kind != ZKW_LQ_LOG) continue;  // refund records are witness only
        const u32 aux_byte = (l7.z >> 16) & 0xffu;
        const bool rw = (l7.w & ZKW_LQ_RW) != 0;
        if (aux_byte == N.storage_aux_byte) {
          if (n_sh >= N.hist_cap) { flags |= 1u; break; }
          st_hist[n_sh++] = lp;
          if (rw) rb_st[n_rs++] = lp;  // n_rs <= number of log records <= per_log
        } else if (aux_byte == N.event_aux_byte || aux_byte == N.l1_aux_byte) {
          if (n_eh >= N.hist_cap) { flags |= 1u; break; }
          ev_hist[n_eh++] = lp;
          rb_ev[n_re++] = lp;
        }
      } else {
        ai++; na--;
        const u32 type = a0.x & 0xffu;
        if (type == ZKW_AUX_FRAME_START) {
          if (n_mk >= N.mark_cap) { flags |= 1u; break; }
          marks[2 * n_mk] = n_rs; marks[2 * n_mk + 1] = n_re;
          n_mk++;
        } else if (type == ZKW_AUX_FRAME_FINISH) {
          if (n_mk == 0) { flags |= 1u; break; }
          n_mk--;
          const bool panicked = ((a0.x >> 24) & 0xffu) != 0;
          if (panicked) {
            const u32 ms = marks[2 * n_mk], me = marks[2 * n_mk + 1];
            if (n_sh + (n_rs - ms) > N.hist_cap || n_eh + (n_re - me) > N.hist_cap) { flags |= 1u; break; }
            while (n_rs > ms) st_hist[n_sh++] = rb_st[--n_rs] | 0x80000000u;
            while (n_re > me) ev_hist[n_eh++] = rb_ev[--n_re] | 0x80000000u;
          }
        }
      }
    }
  }
  outc[0] = n_sh; outc[1] = n_eh; outc[2] = n_re; outc[3] = flags;
}

extern "C" hipError_t zkw_launch_commit(const zkw_fused_table* T, int stage, hipStream_t stream) {
  const u32 wt = T->wave_threads;
  const u32 nq = T->reserved[1] ? (u32)__builtin_popcount(T->reserved[1]) : 1u;  // queues per launch (bucket / chain)
  if (stage == ZKW_COMMIT_STAGE_LEAF) {
    const u32 threads = wt > 1 ? 256 : 1;
    const u32 per_wave_blocks = (T->max_cap + threads - 1) / threads;
    const dim3 grid(per_wave_blocks < 64 ? (per_wave_blocks ? per_wave_blocks : 1) : 64, T->max_waves, T->n);
    hipLaunchKernelGGL(zkw_leaf_kernel<ZKW_QUEUE_CODE_WORDS>, grid, dim3(threads), 0, stream, *T);  // code words only (upload)
  } else if (stage == ZKW_COMMIT_STAGE_BUCKET) {
    hipLaunchKernelGGL(zkw_bucket_kernel, dim3(T->max_waves, T->n, nq), dim3(wt), 0, stream, *T);
  } else if (stage == ZKW_COMMIT_STAGE_CHAIN) {
    hipLaunchKernelGGL(zkw_chain_kernel, dim3(T->max_waves, T->n, nq), dim3(wt), 0, stream, *T);
  } else if (stage == ZKW_COMMIT_STAGE_BLOB_CHUNKS) {
    const u32 threads = wt > 1 ? 64 : 1;
    const u32 max_chunks = (T->max_cap + ZKW_BLOB_CHUNK_WORDS - 1) / ZKW_BLOB_CHUNK_WORDS;  // max_cap = words of the longest blob (upper bound: all words)
    hipLaunchKernelGGL(zkw_blob_chunk_kernel, dim3((max_chunks + threads - 1) / threads ? (max_chunks + threads - 1) / threads : 1, T->n_blobs), dim3(threads), 0, stream, *T);
  } else if (stage == ZKW_COMMIT_STAGE_MIDSTATE) {
    const u32 threads = wt > 1 ? 64 : 1;
    hipLaunchKernelGGL(zkw_midstate_kernel, dim3((T->n_blobs + threads - 1) / threads), dim3(threads), 0, stream, *T);  // n_blobs = number of preimages here
  } else if (stage == ZKW_COMMIT_STAGE_NETSTATE) {
    hipLaunchKernelGGL(zkw_netstate_kernel, dim3(T->max_waves, T->n), dim3(wt), 0, stream, *T);
  } else {
    const u32 threads = wt > 1 ? 64 : 1;
    hipLaunchKernelGGL(zkw_blob_chain_kernel, dim3((T->n_blobs + threads - 1) / threads), dim3(threads), 0, stream, *T);
  }
  return hipGetLastError();
}

// Packs the digests of the committed queues of several batches into the send buffer of the final all-gather:
// dst[batch][row < n_max][q in mask][4] u64 (rows >= n are zero: ragged shards are padded to the largest rank).
// T.p[batch] = that batch's digests [n][ZKW_QUEUE_COUNT][4].
__global__ void zkw_pack_digests_kernel(zkw_fused_table T, uint64_t* dst, u32 n, u32 n_max, u32 mask) {
  const uint64_t* src = (const uint64_t*)T.p[blockIdx.y];
  const u32 nq = (u32)__popcll((unsigned long long)(mask & 7u));
  const u32 total = n_max * nq * 4u;
  uint64_t* out = dst + (size_t)blockIdx.y * total;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const u32 row = i / (nq * 4u), rem = i % (nq * 4u), qi = rem / 4u, e = rem % 4u;
    u32 q = 0, seen = 0;  // the qi-th set bit of the mask
    for (u32 b = 0; b < ZKW_QUEUE_COUNT; b++)
      if ((mask >> b) & 1u) {
        if (seen == qi) q = b;
        seen++;
      }
    out[i] = row < n ? src[((size_t)row * ZKW_QUEUE_COUNT + q) * 4u + e] : 0ull;
  }
}
extern "C" hipError_t zkw_launch_pack_digests(const zkw_fused_table* T, uint64_t* dst, uint32_t n, uint32_t n_max, uint32_t mask, hipStream_t stream) {
  const u32 threads = T->wave_threads > 1 ? 256 : 1;
  const u32 total = n_max * (u32)__builtin_popcount(mask & 7u) * 4u;
  u32 blocks = (total + threads - 1) / threads;
  if (blocks > 64) blocks = 64;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(zkw_pack_digests_kernel, dim3(blocks, T->n), dim3(threads), 0, stream, *T, dst, n, n_max, mask);
  return hipGetLastError();
}
