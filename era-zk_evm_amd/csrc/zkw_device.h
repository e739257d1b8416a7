// Device-side data layout shared by the HIP kernels (zkw_kernels.hip) and the host runtime
// (zkw_runtime.cpp).  See DESIGN.md §"Data layout in HBM".
//
// Mapping: VM instance i  ->  wave w = i / L, lane l = i % L   (L = lanes per wave, 1..64).
// Everything a wave touches every cycle is interleaved across its lanes so that the 64 lanes of
// one load/store instruction hit consecutive 16/32-byte units (coalesced), e.g. the register file
// snapshot of cycle k is `rec[((w*max_cycles + k)*32 + chunk)*L + l]` (16-byte units).
#pragma once
#include <stdint.h>

#include "../../include/zkw.h"

#define ZKW_WAVE 64            /* CDNA wavefront width */
/* ZKW_WIDE: this translation unit runs on real 64-lane waves — the device pass, or the 64-lane CPU emulation of tests/emu
 * (ZKW_EMU_WAVE == 64: every lane a fiber); not defined for the host pass and the one-lane emulation build */
#if defined(__HIP_DEVICE_COMPILE__) || (defined(ZKW_EMU_WAVE) && ZKW_EMU_WAVE > 1)
#define ZKW_WIDE 1
#endif
/* Divergence annotations (no-ops on the device; the 64-lane emulation takes its execution mask from them: zkw_kernels.hip) */
#ifndef ZKW_DIV_IF
#define ZKW_DIV_IF(c) if (c)
#define ZKW_DIV_SCOPE ((void)0)
#define ZKW_LOCKSTEP() ((void)0)
#endif
#define ZKW_WAVES_PER_GROUP 4   /* waves per workgroup of the cycle kernel: one per SIMD, sharing the LDS ISA table */
#define ZKW_KRATE_WORDS 34     /* Keccak rate block (136 B) in dwords */
#define ZKW_EC_SLOTS 12        /* 256-bit values of the ecrecover precompile's point arithmetic kept in the lane's scratch row */
#define ZKW_KROW_WORDS (ZKW_KRATE_WORDS + 8 * ZKW_EC_SLOTS) /* dwords of a lane's scratch row: Keccak block assembly | secp256k1 points */
#define ZKW_REC_CHUNKS 32      /* 512-byte CycleRecord = 32 x 16 B */
#define ZKW_REG_CHUNKS 30      /* 15 registers x 2 halves */

/* packed ISA entry: .x = attributes, .y = price */
#define ZKW_ATTR_OPCODE(a) ((a)&15u)
#define ZKW_ATTR_VARIANT(a) (((a) >> 4) & 15u)
#define ZKW_ATTR_SRC0(a) (((a) >> 8) & 7u)
#define ZKW_ATTR_DST0(a) (((a) >> 11) & 3u)
#define ZKW_ATTR_FLAGS(a) (((a) >> 13) & 3u)
#define ZKW_ATTR_PROPS(a) (((a) >> 15) & 63u)
#define ZKW_ATTR_PACK(op, var, s0, d0, fl, pr) \
  ((uint32_t)(op) | ((uint32_t)(var) << 4) | ((uint32_t)(s0) << 8) | ((uint32_t)(d0) << 11) | ((uint32_t)(fl) << 13) | ((uint32_t)(pr) << 15))
/* bits 21..25: what the cycle kernel's short cycle asks about an instruction, answered once when the table is packed (zkw_short_class) —
 * functions of the fields above, so two entries with equal fields stay equal words */
#define ZKW_ATTR_SHORT_OK (1u << 21)   /* the instruction can run in the short cycle at all (opcode, operand modes, no explicit panic) */
#define ZKW_ATTR_SHORT_MEM (1u << 22)  /* ... and emits memory queries there (a heap access, a code-page operand): needs room in the memory stream */
#define ZKW_ATTR_SHORT_UMA (1u << 23)  /* ... a heap / aux-heap access */
#define ZKW_ATTR_SHORT_CODE (1u << 24) /* ... an ALU instruction whose src0 is a constant from the code page */
#define ZKW_ATTR_SHORT_ALU (1u << 25)  /* ... nop / add / sub / mul / jump / shift / binop */
#define ZKW_ATTR_SHORT_STACK (1u << 26) /* an ALU instruction with a stack operand (src0 read from / dst0 written to the stack, sp moved): short only in
                                          -DZKW_SHORT_STACK builds, and never together with ZKW_ATTR_SHORT_OK */
static inline uint32_t zkw_short_class(uint32_t op, uint32_t var, uint32_t s0, uint32_t d0, uint32_t pr) {
  const int alu = op == ZKW_OP_NOP || op == ZKW_OP_ADD || op == ZKW_OP_SUB || op == ZKW_OP_MUL || op == ZKW_OP_JUMP || op == ZKW_OP_SHIFT || op == ZKW_OP_BINOP;
  const int uma = op == ZKW_OP_UMA && var <= ZKW_UMA_AUX_WRITE && !(pr & ZKW_PROP_SWAP);
  const int code = s0 == ZKW_MODE_CODE && alu && op != ZKW_OP_NOP;
  const int ok = (alu || uma) && d0 == ZKW_MODE_REG && (s0 == ZKW_MODE_REG || s0 == ZKW_MODE_IMM || code) && !(pr & ZKW_PROP_EXPLICIT_PANIC);
  const int s0_stack = s0 == ZKW_MODE_STACK_PP || s0 == ZKW_MODE_STACK_OFF || s0 == ZKW_MODE_STACK_ABS;
  const int d0_stack = d0 == ZKW_MODE_STACK_PP || d0 == ZKW_MODE_STACK_OFF || d0 == ZKW_MODE_STACK_ABS;
  const int stack = alu && (s0_stack || d0_stack) && (s0_stack || s0 == ZKW_MODE_REG || s0 == ZKW_MODE_IMM) && (d0_stack || d0 == ZKW_MODE_REG) &&
                    !(op == ZKW_OP_JUMP && d0_stack) && !(pr & ZKW_PROP_EXPLICIT_PANIC);
  return (ok ? ZKW_ATTR_SHORT_OK : 0u) | (ok && (uma || code) ? ZKW_ATTR_SHORT_MEM : 0u) | (uma ? ZKW_ATTR_SHORT_UMA : 0u) | (code ? ZKW_ATTR_SHORT_CODE : 0u) |
         (alu ? ZKW_ATTR_SHORT_ALU : 0u) | (stack ? ZKW_ATTR_SHORT_STACK : 0u);
}

/* callstack entry as kept on device: the ABI struct + what the device needs to re-enter the frame */
typedef struct zkw_dev_entry {
  zkw_callstack_entry e; /* 112 B */
  uint32_t code_blob;    /* blob backing e.code_page                                  */
  uint32_t frame_slot;   /* memory arena slot of the far frame this entry lives in     */
  uint32_t journal_mark; /* storage journal length when the frame started              */
  uint32_t reserved;
} zkw_dev_entry; /* 128 B */

/* per-instance scalar state (everything of VmLocalState that is not the register file or the
 * callstack), plus run bookkeeping.  Loaded into VGPRs at kernel start, stored at kernel end. */
typedef struct zkw_dev_scalars {
  uint32_t prev_code_word[8];
  uint32_t ctx_u128_reg[4];
  uint32_t ptr_bitmap;          /* bit i = registers[i].is_pointer */
  uint32_t flags;               /* bits 0..2 lt/eq/gt, bit 3 pending_exception */
  uint32_t prev_code_page;
  uint32_t timestamp;
  uint32_t cycle_counter;       /* monotonic_cycle_counter */
  uint32_t spent_pubdata;
  uint32_t memory_page_counter;
  uint32_t absolute_execution_step;
  uint32_t ergs_per_pubdata;
  uint32_t tx_number;
  uint32_t prev_super_pc;
  uint32_t depth;               /* callstack.inner.len() */
  uint32_t status;              /* ZKW_STATUS_* */
  uint32_t n_cycles;            /* cycles executed since reset */
  uint32_t first_dynamic_page;  /* memory_page_counter at reset: far call j gets base page first + 8 j */
  uint32_t n_initial_slots;     /* arena slots taken by the frames alive at reset */
  uint32_t next_slot;           /* next free arena slot */
  uint32_t journal_len;         /* storage journal length */
  uint32_t n_history;           /* decommitter history length */
  uint32_t reserved[1];
} zkw_dev_scalars; /* 32 x 4 = 128 B */

/* per (instance, arena slot): lazily-zeroed pages: words >= hwm read as zero */
typedef struct zkw_dev_frame_meta {
  uint32_t base_page;
  uint32_t stack_hwm;
  uint32_t heap_hwm;
  uint32_t aux_hwm;
} zkw_dev_frame_meta;

typedef struct zkw_dev_storage_entry {
  uint32_t key[8];
  uint32_t value[8];
  uint32_t address[5];
  uint32_t shard_state; /* bits 0-7 shard id, bit 8 occupied, bit 9 warm */
  uint32_t reserved[2];
} zkw_dev_storage_entry; /* 96 B */

typedef struct zkw_dev_journal_entry {
  uint32_t old_value[8];
  uint32_t slot;
  uint32_t reserved[3];
} zkw_dev_journal_entry; /* 48 B */

typedef struct zkw_dev_preimage {
  uint32_t hash[8];
  uint32_t blob;
  uint32_t reserved[3];
} zkw_dev_preimage; /* 48 B */

/* SimpleDecommitter's history (decommitter.rs:38-47: a map hash -> page, one row per fresh decommit, never bounded): one
 * row per (hash -> blob) pair the batch knows, indexed by the pair — a lookup is one load, and the decommits of a run are
 * capped by nothing but the number of known code hashes (an unknown hash is the reference's Err) */
typedef struct zkw_dev_history {
  uint32_t valid; /* 1: this code hash was decommitted since the reset ... */
  uint32_t page;  /* ... into this page */
} zkw_dev_history;

/* The parameter block of a batch lives in device memory (written once at upload) and is read through the
 * constant address space, so that its fields are scalar (s_load) loads exactly like kernel arguments while one
 * launch can cover several batches (grid.y = batch).  The CPU emulation build of tests/emu has no address spaces. */
#ifdef __HIP_DEVICE_COMPILE__
#define ZKW_CONST_AS __attribute__((address_space(4)))
#else
#define ZKW_CONST_AS
#endif
/* slot of a code hash in zkw_kparams.pre_index (host: zkw_batch_upload; device: op_far_call) */
#if defined(__HIP__)
#define ZKW_HD __host__ __device__
#else
#define ZKW_HD
#endif
ZKW_HD static inline uint32_t zkw_pre_hash(const uint32_t h[8]) {
  uint32_t x = h[0] ^ (h[1] * 0x9e3779b9u) ^ (h[2] * 0x85ebca6bu);
  x ^= x >> 15;
  return x * 0xc2b2ae35u;
}
#define ZKW_MAX_FUSED 256 /* batches per fused launch (the by-value tables stay under the 4 KB kernel-argument segment) */

/* kernel parameter block */
typedef struct zkw_kparams {
  uint32_t n_instances;
  uint32_t L;          /* lanes per wave */
  uint32_t n_waves;
  uint32_t max_cycles; /* limits.max_cycles (record capacity per instance) */
  uint32_t reserved1[2];
  uint32_t F, D, S, H, A; /* max_far_frames, max_callstack_depth, stack/heap/aux words */
  uint32_t storage_slots, storage_journal;
  uint32_t cap_mem, cap_log, cap_aux; /* stream capacity per wave (records) */
  uint32_t cap_delta;  /* register-delta capacity per wave (32-B values) */
  uint32_t reserved3;
  uint32_t n_blobs, n_preimages;
  uint32_t wave_threads; /* hardware wave width (64 on gfx950; 1 in the CPU emulation build of tests/emu) */
  uint32_t waves_per_group; /* unused: the waves per workgroup are a property of the launch (zkw_launch_args) */
  uint32_t reserved0;
  uint32_t reserved2;
  zkw_isa_consts consts;
  zkw_block_properties props;
  const uint2* isa;            /* [2048] packed */
  /* state */
  uint4* regs;                 /* [n_waves][30][L]                      */
  uint32_t* krow;              /* [n_waves][34][L] Keccak block assembly rows */
  zkw_dev_scalars* scalars;    /* [n_instances]                          */
  zkw_dev_entry* callstack;    /* [n_instances][D + 1]                   */
  zkw_dev_frame_meta* frames;  /* [n_instances][F]                       */
  uint4* stack_vals;           /* [n_waves][F][S][2][L]                  */
  uint8_t* stack_ptrs;         /* [n_waves][F][S][L]                     */
  uint4* heap;                 /* [n_waves][F][H][2][L]                  */
  uint4* aux_heap;             /* [n_waves][F][A][2][L]                  */
  zkw_dev_storage_entry* storage;  /* [n_instances][storage_slots]       */
  zkw_dev_journal_entry* journal;  /* [n_instances][storage_journal]     */
  zkw_dev_history* history;        /* [n_instances][hist_pitch]: row p = preimage p (zkw_dev_history) */
  /* read-only inputs */
  const uint4* blob_words;     /* all code blobs, 2 x uint4 per word     */
  const uint2* blob_dir;       /* [n_blobs] (first word, n_words)        */
  const zkw_dev_preimage* preimages; /* [n_preimages]                    */
  /* outputs */
  uint4* tails;                /* [n_waves][max_cycles][L]: 16 B per executed cycle: pointer bitmap, flags, pc, sp, ergs, event counts + the 16-bit delta mask */
  uint4* deltas;               /* [n_waves][2][cap_delta]: 32-B values of the registers a cycle wrote (mask bits 0..14) and of the tail's slow half (bit 15: heap bound, aux bound, depth), dense per wave, as two planes (low / high 16 B) */
  uint32_t* wave_cycles;       /* [n_waves] wave-cycles run since the reset */
  uint32_t* heap_dirty;        /* [n_waves][ceil(heap_image_words / 32)][L]: words of the heap image overwritten since the reset */
  const uint4* regs0;          /* pristine register files / scalars: what a wave starts from in its first launch after a  */
  const zkw_dev_scalars* scalars0; /* reset (wave_cycles == 0) — the reset does not copy them into the working buffers   */
  uint32_t* storage_dirty;     /* [n_instances][ceil(storage_slots / 32)]: one bit per storage-table slot written since the reset */
  uint32_t heap_image_words;   /* words of the uploaded heap image (frame slot 0) */
  uint32_t hist_pitch;         /* rows of `history` per instance = max(1, n_preimages) */
  const uint32_t* pre_index;   /* [pre_mask + 1] open addressing over the code hashes: 0 = empty, else preimage index + 1 (zkw_pre_hash) */
  uint32_t pre_mask;
  uint32_t reserved5;
  uint4* mem_stream;           /* [n_waves][3][cap_mem]: planes header | value low | value high of the 48-byte zkw_mem_query */
  uint4* log_stream;           /* [n_waves][cap_log][8]                  */
  uint4* aux_stream;           /* [n_waves][cap_aux][16]                 */
  /* decommit-queue commitment, chained by the cycle kernel itself at every decommit (zkw_commit.hip spec) */
  const uint64_t* commit_rc;    /* [ZKW_GL_RC_COUNT] round constants */
  const uint64_t* midstates;    /* [n_preimages][12] sponge state after absorbing a code hash */
  const uint64_t* blob_digests; /* [n_blobs][4] */
  uint64_t* commit_out;         /* [n_instances][ZKW_QUEUE_COUNT][4]: the DECOMMIT slot is the running tail */
  uint32_t* dq_count;           /* [n_instances] decommit-queue length so far */
  uint64_t* dq_prev;            /* [n_instances][4] the tail before the decommit chained last: what a cycle that fails AFTER its decommit restores */
  uint32_t* dir;               /* [n_waves][max_cycles + 1][4] (mem, log, aux cursors at cycle start) */
  uint32_t* cursors;           /* [n_waves][4] persistent stream cursors: mem, log, aux, register deltas */
  const zkw_dev_entry* callstack0; /* [n_instances][D + 1] pristine callstack (zkw_expand_kernel: the pc / memory bounds an instance started with) */
} zkw_kparams;
#define ZKW_KP const zkw_kparams ZKW_CONST_AS&

/* by-value arguments of one (possibly fused) launch of the cycle kernel: grid.y = batch */
#define ZKW_DQ_HELPER (1u << 27)     /* zkw_launch_args.debug_flags: decommits are posted to the workgroup's helper wave */
#define ZKW_NO_DQ_HELPER (1u << 28)  /* ZKW_OPT_DEBUG_FLAGS: never launch helper waves (A/B) */
#define ZKW_DQ_HELPER_BYTES 1552u    /* LDS per cycle wave: 16 B of counters + a ring of 2 x [3][64] dwords */
#define ZKW_NO_PREFETCH (1u << 30)   /* ZKW_OPT_DEBUG_FLAGS: no heap prefetch for the next instruction (A/B) */
#define ZKW_KECCAK_HELPER (1u << 29) /* zkw_launch_args.debug_flags: every cycle wave has helper waves of its own that serve its keccak256
                                        calls lane-parallel (one lane per half state word: zkw_kh_helper) — batches of thin waves (<= 8 lanes) only */
#define ZKW_KH_MAX_LANES 8u          /* ... which is the number of request rows in the mailbox of a cycle wave */
#define ZKW_KH_BYTES (ZKW_KH_MAX_LANES * 64u) /* LDS per cycle wave: one 16-dword row per lane (request in, digest out) */
#define ZKW_MAX_WAVES_PER_GROUP 8 /* a CU holds 8 waves of the cycle kernel (256 registers: two per SIMD) */
typedef struct zkw_launch_args {
  const zkw_kparams* kp[ZKW_MAX_FUSED]; /* device copies of the parameter blocks */
  /* The waves of all batches of the launch are numbered through (batch 0's first): workgroup j runs waves
   * j * waves_per_group ..., so that the number of waves per workgroup can be chosen for the launch as a whole (1280 waves
   * = 256 workgroups of 5: one per CU) and not per batch.  wave_base[b] = first wave of batch b, wave_base[n_batches] = all;
   * uniform_waves != 0: every batch has that many waves (batch = wave / uniform_waves, no search). */
  uint32_t wave_base[ZKW_MAX_FUSED + 1];
  uint32_t uniform_waves;
  uint32_t helpers;   /* helper waves per workgroup.  1: one more wave that chains the decommit-queue commitment for the cycle waves
                         (ZKW_DQ_HELPER in debug_flags; only when the CUs have a wave slot to spare); a multiple k of waves_per_group with
                         ZKW_KECCAK_HELPER: helpers h * k .. h * k + k - 1 serve cycle wave h (keccak256 calls of its lanes r with r % k = the
                         helper's index; the first also its decommits under ZKW_DQ_HELPER) */
  uint32_t lds_sink;  /* byte offset, in the dynamic LDS of a workgroup, of the 256 bytes the prefetches of the cycle kernel land in (set by the launcher) */
  uint32_t n_batches;
  uint32_t run_cycles;
  uint32_t debug_flags; /* profiling ablations / test hooks only (ZKW_DEBUG_FLAGS): 1 = no CycleRecord stores, 2 = no stream stores, 4 = one lane per group */
  uint32_t max_waves, max_L, wave_threads, waves_per_group; /* launch geometry (host side of the launcher) */
} zkw_launch_args;

/* by-value argument of the reset / commitment kernels: device copies of per-batch parameter structs, one batch
 * per grid.y (leaf kernel: grid.z) */
typedef struct zkw_fused_table {
  const void* p[ZKW_MAX_FUSED];
  uint32_t n;
  uint32_t max_waves;    /* launch geometry (host side of the launchers): max over the batches */
  uint32_t max_cap;      /* leaf kernel: upper bound of records per wave */
  uint32_t wave_threads;
  uint32_t n_blobs;      /* blob-chain stage only */
  uint32_t reserved[3];  /* [0]: leaf stage: the queue (ZKW_QUEUE_* / ZKW_QUEUE_CODE_WORDS) of the blocks in the table;
                            [1]: reset kernel: 1 = copy the whole heap image (first reset after an upload); [2]: reset kernel: parts left out (ablation) */
} zkw_fused_table;

/* zkw_reset_kernel: working state := pristine images, one launch */
typedef struct zkw_reset_params {
  uint4* dst[5];
  const uint4* src[5];
  uint32_t n16[5];          /* 16-byte units per buffer */
  uint32_t cs_row16;         /* callstack ([2]): 16-byte units to restore per instance (entries 0..initial depth) ... */
  uint32_t cs_pitch16;       /* ... out of this many per instance; n16[2] = n_instances * cs_row16 */
  uint4* heap_dst;           /* working heap arena */
  const uint4* heap_src;     /* [n_waves][heap_image_words][2][L] */
  uint32_t heap_row16;       /* 16-byte units per wave row of the image */
  uint32_t heap_pitch16;     /* 16-byte units between wave rows in the arena */
  uint32_t n_waves;
  uint32_t* cursors;         /* [n_waves][4] */
  uint32_t* wave_cycles;     /* [n_waves] */
  uint32_t* heap_dirty;      /* [n_waves][ceil(image_words / 32)][L] */
  uint32_t image_words;      /* words of the heap image */
  uint32_t L;
  uint64_t* commit_out;      /* [n_instances][ZKW_QUEUE_COUNT][4]: the running decommit-queue tails are zeroed */
  uint32_t* dq_count;        /* [n_instances] */
  uint32_t n_instances;
  uint32_t storage_slots;    /* slots per instance of the storage table ([4]: restored per dirty slot after the first reset) */
  uint32_t* storage_dirty;   /* [n_instances][ceil(storage_slots / 32)] */
  uint4* history;            /* decommit history rows, cleared by every reset ... */
  uint32_t history16;        /* ... 16-byte units of it */
  uint32_t reserved0;
} zkw_reset_params;
