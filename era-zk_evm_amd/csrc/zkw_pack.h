// Link format of a delivered step (zkw_pack_kernel -> pinned host memory -> the host rebuild of zkw_runtime.cpp).
//
// What the reference hands its VmWitnessTracer lives on the HOST (all ten callbacks of witness_trace/mod.rs:11-72 run on the
// CPU), so every trace has to cross PCIe.  The cycle kernel leaves its streams in HBM at fixed capacities per wave; the pack
// kernel writes the USED extents of one step — every wave of every batch of a fused group — into ONE contiguous block, and it
// writes that block straight into pinned host memory (the kernel's stores go over the link: 55 GB/s of the 57 GB/s a
// hipMemcpyAsync of the same bytes reaches on these boxes, profiles/r08_pcie_probe.txt — and no staging copy, no sizes the
// host would have to know before it can enqueue a copy).  The block is also a denser encoding than the device streams:
//   memory-query header  16 B -> 12 B   three u32 planes (page | index | lane, seq, meta, low byte of the timestamp: a query's
//                                       timestamp is the cycle's + 0..3, mod.rs:220-231, and the rebuild knows the cycle's)
//   code-word queries    no value       MemoryType::Code reads (cycle.rs:76-81, mem_ops.rs:100-110) return words of a code page
//                                       the host holds (the blobs it uploaded): the value planes carry the other queries only
//   memory READS         no value       (version 2, ZKW_PACK_NO_READ_VALUES in the header's flags) what a read of a stack / heap / aux-heap page
//                                       returns follows from what was written there before: the host rebuild keeps a shadow of every
//                                       page a lane touches — the heap image it staged, zero elsewhere (SimpleMemory: fresh pages are
//                                       zero-filled, reference_impls/memory.rs:15-148, 413-521), every write query of the stream applied
//                                       in order — and fills `MemoryQuery.value` of the reads itself; only WRITES carry their value.
//                                       Used when the host holds the heap images the step ran on: a copy made at upload, or the pinned
//                                       staging buffer a restage handed them over in (kept by the step's ticket, ZKW_OPT_STAGING_BUFFERS);
//                                       a batch restaged through a ring of ONE buffer keeps its read values
//   memory-query page    implied        (version 2, ZKW_PACK_IMPLIED_PAGES) the page of a stack / heap / aux-heap / code query of the VM itself is
//                                       base + 1 / + 2 / + 3 / the code page of the frame that is current in that cycle (execution_stack.rs:67-81,
//                                       mem_ops.rs:51-121, uma.rs:100-135): the rebuild tracks the frames (initial callstack + the FRAME_START /
//                                       FRAME_FINISH events of the aux stream) and fills it in; only queries that NAME a page travel with it —
//                                       fat-pointer reads and the reads / writes of precompiles.  The header of the others is 8 bytes.
//   aux events           256 B -> used  FRAME_START 240 B, DECOMMIT 64 B, COLD_STATE 48 B, FRAME_FINISH 16 B
// All offsets and sizes are in 16-byte units from the start of the block.
//
//   [0, 4)                      zkw_pack_header (written by the host at submit; `used_units` / `overflow` land behind the kernel)
//   [4, ...)                    zkw_pack_batch   x n_batches   (host)
//   [...]                       zkw_pack_wave    x n_waves     (kernel: one entry per wave of the step, numbered through the batches)
//   per batch, when the step was packed `with_instances`:
//     scalars  [n_instances] x 8 units      zkw_dev_scalars (status, cycle counts, the scalar half of the final VmLocalState)
//     regs     [n_waves][30][L]             the final register files, device layout
//     entries  [n_instances] x 8 units      callstack.current of the final state (zkw_dev_entry)
//   wave data, allocated in the order the workgroups get to them (zkw_pack_wave.off):
//     dir      (max_cyc + 1)                stream cursors at every cycle start
//     tails    max_cyc x L x 16 B — or, slim: planes x | y | z of ceil4(max_cyc x L) units each (ptr bitmap, flags, low delta mask | pc, sp |
//              ergs) and one plane of ceil16(max_cyc x L) units with the high byte of the delta mask — or, as differences
//              (ZKW_PACK_DELTA_TAILS): one plane of u32 (zkw_pack_tail_word: delta mask | ergs spent in the cycle | flags | "pc went on by
//              one and sp stayed" | "the pointer bitmap stayed"), then three lists of u32 with what was not as predicted, each in
//              lane-cycle order: ptr bitmap | flags, pc | sp, ergs (n_tx, n_ty, n_tz entries; every lane's first cycle is in all three)
//     deltas   n_delta (low plane), n_delta (high plane) — or sparse (ZKW_PACK_SPARSE_DELTAS, zkw_pack_delta_units): bit planes, bytes 0..7 of
//              every delta, bytes 8..15 of those that have any, bytes 16..31 of those that have any (addresses, counters, lengths and
//              flags are most of what a VM writes to its registers: half of the deltas of the headline tape fit 16 bytes, 0.4 fit 8)
//     mem      the page list (ceil4(n_page) units of u32: the pages of the queries that carry one, zkw_pack_has_page, in stream
//              order — every query when pages are not implied), 2 planes of ceil4(n_mem) u32 (index | misc), then n_val (value
//              low), n_val (value high): the values of
//              the queries that carry one (zkw_pack_has_value: no Code reads; under ZKW_PACK_NO_READ_VALUES no reads at all), in
//              stream order
//     log      n_log x 8
//     aux      aux_units (records back to back, each as long as its type uses)
#pragma once
#include <stdint.h>

#include "zkw_device.h"

#define ZKW_PACK_MAGIC 0x50574b5au /* "ZKWP" */
#define ZKW_PACK_VERSION 2u
#define ZKW_PACK_NO_READ_VALUES 1u /* zkw_pack_header.flags / zkw_pack_args.flags: the value planes hold the values of WRITES only */
#define ZKW_PACK_IMPLIED_PAGES 2u  /* ... the page list holds the pages of fat-pointer and precompile queries only */
#define ZKW_PACK_SLIM_TAILS 4u     /* ... the record tails travel as 13 bytes (three u32 planes + one byte plane) instead of 16: the three
                                      event counts of a cycle are the numbers of its queries in the streams, which the rebuild counts anyway */
#define ZKW_PACK_SPARSE_DELTAS 8u /* ... a register delta travels as its low 8 bytes + bytes 8..15 if any is set + bytes 16..31 if any is set (two bit planes say which) */
#define ZKW_PACK_DELTA_TAILS 16u   /* ... a record tail travels as ONE u32 (delta mask | ergs spent | flags | "pc, sp, pointer bitmap as predicted") and the words
                                      that were not as predicted in three lists */
#define ZKW_PACK_HEADER_UNITS 4u
#define ZKW_PACK_BATCH_UNITS 2u
#define ZKW_PACK_WAVE_UNITS 4u
#define ZKW_PACK_THREADS 256u
#define ZKW_PACK_MAX 256u /* batches per launch */

typedef struct zkw_pack_header { /* 64 B */
  uint32_t magic, version, n_batches, n_waves;
  uint32_t fixed_units;    /* header + tables + instance sections: where the wave data starts */
  uint32_t with_instances; /* 1: the per-instance sections are present */
  uint32_t used_units;     /* total units of the block (copied behind the kernel from its allocation cursor) */
  uint32_t overflow;       /* != 0: the block did not fit the slot: waves without data have off == 0 */
  uint32_t flags;          /* ZKW_PACK_* */
  uint32_t reserved[7];
} zkw_pack_header;

typedef struct zkw_pack_batch { /* 32 B */
  uint32_t n_instances, L, n_waves, first_wave; /* first_wave: index of its wave 0 in the wave table */
  uint32_t scalars_off, regs_off, entries_off;  /* unit offsets of the instance sections (0 without them) */
  uint32_t max_cycles;
} zkw_pack_batch;

typedef struct zkw_pack_wave { /* 64 B */
  uint32_t off;       /* first unit of the wave's data; 0 = not packed (overflow) */
  uint32_t max_cyc;   /* wave-cycles run since the reset */
  uint32_t n_delta, n_mem, n_val, n_log, n_aux, aux_units;
  uint32_t units;     /* units of the wave's data */
  uint32_t n_page;    /* entries of the page list */
  uint32_t n_d1, n_d2; /* sparse deltas: entries of the plane of bytes 8..15, of the plane of bytes 16..31 */
  uint32_t n_tx, n_ty, n_tz; /* delta tails: entries of the three lists (pointer bitmap | flags, pc | sp, ergs) */
  uint32_t reserved;
} zkw_pack_wave;

/* units a record of the aux stream uses, by type (the kernel writes only those: zkw_kernels.hip start_frame / op_far_call) */
ZKW_HD static inline uint32_t zkw_aux_used_units(uint32_t type) {
  return type == ZKW_AUX_FRAME_START ? 15u : type == ZKW_AUX_DECOMMIT ? 4u : type == ZKW_AUX_COLD_STATE ? 3u : 1u;
}
ZKW_HD static inline uint32_t zkw_ceil4(uint32_t n) { return (n + 3u) >> 2; }
/* what the size of a wave's data follows from (the wave's table entry holds the same numbers) */
typedef struct zkw_pack_counts {
  uint32_t max_cyc, L, n_delta, n_mem, n_page, n_val, n_log, aux_units;
  uint32_t n_d1, n_d2;       /* ZKW_PACK_SPARSE_DELTAS */
  uint32_t n_tx, n_ty, n_tz; /* ZKW_PACK_DELTA_TAILS */
} zkw_pack_counts;
ZKW_HD static inline uint64_t zkw_ceil2_64(uint64_t n) { return (n + 1ull) >> 1; }
ZKW_HD static inline uint64_t zkw_ceil4_64(uint64_t n) { return (n + 3ull) >> 2; }
/* units of the tails of n_t lane-cycles: 16 bytes each; slim: three u32 planes + one byte plane; as differences: one u32 plane + the three lists */
ZKW_HD static inline uint64_t zkw_pack_tail_units(uint64_t n_t, const zkw_pack_counts* c, uint32_t flags) {
  if (flags & ZKW_PACK_DELTA_TAILS) return zkw_ceil4_64(n_t) + zkw_ceil4_64(c->n_tx) + zkw_ceil4_64(c->n_ty) + zkw_ceil4_64(c->n_tz);
  return (flags & ZKW_PACK_SLIM_TAILS) ? 3ull * ((n_t + 3ull) >> 2) + ((n_t + 15ull) >> 4) : n_t;
}
/* units of the register deltas.  Sparse: per 256 deltas 4 units of bit planes (below), then the plane of bytes 0..7 (u64 each), the plane of
 * bytes 8..15 of the deltas that have any (u64 each), the plane of bytes 16..31 of the deltas that have any (16 B each).
 * Bit planes: delta i = 256 * blk + 4 * lane + j is bit `lane` of u64 word 8 * blk + j ("bytes 8..15 present") and of word 8 * blk + 4 + j
 * ("bytes 16..31 present"): what four ballots of a 64-lane wave that holds four consecutive deltas per lane produce. */
ZKW_HD static inline uint64_t zkw_pack_delta_units(const zkw_pack_counts* c, uint32_t flags) {
  if (!(flags & ZKW_PACK_SPARSE_DELTAS)) return 2ull * c->n_delta;
  return 4ull * (((uint64_t)c->n_delta + 255ull) >> 8) + zkw_ceil2_64(c->n_delta) + zkw_ceil2_64(c->n_d1) + c->n_d2;
}
/* units of a wave's data */
ZKW_HD static inline uint64_t zkw_pack_wave_units(const zkw_pack_counts* c, uint32_t flags) {
  return (uint64_t)(c->max_cyc + 1u) + zkw_pack_tail_units((uint64_t)c->max_cyc * c->L, c, flags) + zkw_pack_delta_units(c, flags) + zkw_ceil4(c->n_page) +
         2ull * zkw_ceil4(c->n_mem) + 2ull * c->n_val + 8ull * c->n_log + c->aux_units;
}
/* ... at most, whatever the flags, with streams of these capacities */
ZKW_HD static inline uint64_t zkw_pack_wave_units_worst(uint32_t max_cyc, uint32_t L, uint32_t cap_delta, uint32_t cap_mem, uint32_t cap_log, uint32_t cap_aux) {
  const uint64_t n_t = (uint64_t)max_cyc * L;
  return (uint64_t)(max_cyc + 1u) + (n_t + 16ull) + (2ull * cap_delta + 4ull * (((uint64_t)cap_delta + 255ull) >> 8) + 2ull) + zkw_ceil4(cap_mem) + 2ull * zkw_ceil4(cap_mem) +
         2ull * cap_mem + 8ull * cap_log + 16ull * cap_aux;
}

/* The u32 a record tail travels as under ZKW_PACK_DELTA_TAILS.  `t`: the tail as the cycle kernel stored it (x = ptr bitmap | flags << 16 | low delta
 * mask << 24, y = pc | sp << 16, z = ergs, w = event counts | high delta mask << 24); `prev`: the tail of the lane's cycle before; `first`: there is
 * none (the lane's first cycle since the reset: everything travels).
 *   bits 0..15 delta mask   16..23 ergs spent in the cycle, 255 = see the ergs list (a refund, a far call's stipend, a price >= 255)
 *   24..27 flags (lt | eq | gt | pending exception)   28 pc = previous pc + 1 and sp unchanged, else see the pc | sp list
 *   30 pointer bitmap unchanged (and the flags fit their four bits), else see the ptr bitmap | flags list */
#define ZKW_TW_Y_PRED (1u << 28)
#define ZKW_TW_X_SAME (1u << 30)
ZKW_HD static inline uint32_t zkw_pack_tail_word(uint4 t, uint4 prev, bool first) {
  const uint32_t dmask = (t.x >> 24) | ((t.w >> 24) << 8);
  const uint32_t spent = prev.z - t.z;
  const bool z_in = !first && spent < 255u;
  const bool y_pred = !first && (t.y & 0xffffu) == ((prev.y + 1u) & 0xffffu) && (t.y >> 16) == (prev.y >> 16);
  const bool x_same = !first && (t.x & 0xffffu) == (prev.x & 0xffffu) && ((t.x >> 16) & 0xf0u) == 0;
  return dmask | ((z_in ? spent : 255u) << 16) | (((t.x >> 16) & 0xfu) << 24) | (y_pred ? ZKW_TW_Y_PRED : 0u) | (x_same ? ZKW_TW_X_SAME : 0u);
}
ZKW_HD static inline bool zkw_tw_has_x(uint32_t w) { return !(w & ZKW_TW_X_SAME); }
ZKW_HD static inline bool zkw_tw_has_y(uint32_t w) { return !(w & ZKW_TW_Y_PRED); }
ZKW_HD static inline bool zkw_tw_has_z(uint32_t w) { return ((w >> 16) & 0xffu) == 255u; }

/* by-value arguments of one launch of the pack kernel */
typedef struct zkw_pack_args {
  const zkw_kparams* kp[ZKW_PACK_MAX];
  uint32_t wave_base[ZKW_PACK_MAX + 1]; /* the waves of the launch numbered through its batches */
  const zkw_pack_batch* batches;        /* device copy of the block's batch table (the instance-section offsets; the by-value
                                           tables above already fill most of the 4 KB kernel-argument segment) */
  uint4* dst;           /* the block (device-visible address of the pinned slot, or device memory) */
  uint32_t* state;      /* device: [0] allocation cursor (units; starts at fixed_units), [1] overflow flag */
  uint32_t dst_units;   /* capacity of the block */
  uint32_t n_batches;
  uint32_t wave_table;  /* unit offset of the wave table */
  uint32_t with_instances;
  uint32_t only_wave;   /* 0xffffffff: every wave; else the one wave (of the one batch) to pack — the on-demand path of zkw_batch_get_instance_trace */
  uint32_t flags;       /* ZKW_PACK_NO_READ_VALUES */
} zkw_pack_args;
/* does memory query `hdr_w` (word 3 of its header: lane | seq << 8 | meta << 16) carry its value on the link? */
/* ... and its page?  (with implied pages: only what names one — a fat-pointer read, a precompile's read or write) */
ZKW_HD static inline bool zkw_pack_has_page(uint32_t hdr_w, uint32_t flags) {
  const uint32_t meta = (hdr_w >> 16) & 0xffu;
  return !(flags & ZKW_PACK_IMPLIED_PAGES) || (meta >> ZKW_MQ_KIND_SHIFT) != 0 || (meta & ZKW_MQ_TYPE_MASK) == ZKW_MEM_FAT_PTR;
}
ZKW_HD static inline bool zkw_pack_has_value(uint32_t hdr_w, uint32_t flags) {
  const uint32_t meta = hdr_w >> 16;
  if ((meta & ZKW_MQ_TYPE_MASK) == ZKW_MEM_CODE) return false;
  return !(flags & ZKW_PACK_NO_READ_VALUES) || (meta & ZKW_MQ_RW) != 0;
}

/* zkw_restage_kernel: fresh inputs of an uploaded batch, brought into the device layouts ON the device.  The host hands over
 * what the C ABI takes — VmLocalStates [n] and heap images [n][image_words] u256, instance-major, through pinned staging and
 * one H2D copy each — and the kernel writes the pristine images: the lane-interleaved register files ([wave][30][L]) and heap
 * image ([wave][word][2][L]), the scalars, the `current` row of the callstack.  (Formatted on the host the interleave is a
 * 33 MB scatter per 4096-instance batch: several milliseconds of one core against the 0.65 ms its H2D copy takes.) */
typedef struct zkw_restage_params {
  const zkw_vm_local_state* states; /* device copy of the caller's states [n_instances] */
  const uint4* heaps;               /* device copy of the heap images [n_instances][image_words][2] (NULL: heaps unchanged) */
  uint4* regs0;                     /* [n_waves][30][L] */
  zkw_dev_scalars* scalars0;        /* [n_instances] */
  zkw_dev_entry* callstack0;        /* [n_instances][D + 1] */
  uint4* heap0;                     /* [n_waves][image_words][2][L] */
  uint32_t n_instances, L, n_waves, D, image_words;
  uint32_t F;                       /* arena slots per instance (rows of frames0) */
  zkw_dev_frame_meta* frames0;      /* [n_instances][F]: with heaps, slot 0's heap mark becomes image_words — the uploaded images may
                                       have been shorter per instance (words at and beyond the mark read as zero) */
} zkw_restage_params;
