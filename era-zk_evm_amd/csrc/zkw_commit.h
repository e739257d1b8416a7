// Parameters of the queue-commitment kernels (zkw_commit.hip) — see DESIGN.md §commitments.
#pragma once
#include <stdint.h>

#include "zkw_device.h"

#define ZKW_GL_RC_COUNT (4 * 12 + 22 + 4 * 12)
#define ZKW_LEAF_MEM 1
#define ZKW_LEAF_LOG 2
#define ZKW_LEAF_DECOMMIT 3
#define ZKW_LEAF_CODE_WORD 4
#define ZKW_QUEUE_CODE_WORDS 3 /* pseudo queue: leaves of the code blobs */
#define ZKW_QUEUE_ID_BLOB 0xB10Bu      /* chains inside a 64-word chunk of a code blob */
#define ZKW_QUEUE_ID_BLOB_TOP 0xB10Cu  /* chain over the chunk tails of a blob */
#define ZKW_BLOB_CHUNK_WORDS 64
#define ZKW_COMMIT_STAGE_LEAF 0
#define ZKW_COMMIT_STAGE_BUCKET 1
#define ZKW_COMMIT_STAGE_CHAIN 2
#define ZKW_COMMIT_STAGE_BLOB_CHAIN 3
#define ZKW_COMMIT_STAGE_NETSTATE 4
#define ZKW_COMMIT_STAGE_MIDSTATE 5
#define ZKW_COMMIT_STAGE_BLOB_CHUNKS 6

typedef struct zkw_commit_params {
  uint32_t n_instances, L, n_waves, max_cycles, wave_threads;
  uint32_t queue;            /* ZKW_QUEUE_* or ZKW_QUEUE_CODE_WORDS */
  uint32_t cap;              /* stream capacity per wave (records) */
  uint32_t per_instance_cap; /* idx capacity per instance */
  uint32_t n_blobs;
  uint32_t n_override;       /* != 0: number of records instead of cursors[] (code-word leaves) */
  uint32_t aux_type_mask;    /* bucket pass over the aux stream: bit t keeps events of type t (0 = DECOMMIT only) */
  uint32_t pooled;           /* bucket / chain: 1 = the index lists of a wave share its `cap` entries (offs[]): an instance may hold more
                                than per_instance_cap records as long as its wave's stream held them (the commitments); 0 = fixed
                                per-instance lists, counts clamped to per_instance_cap (the netting pass reports the overflow) */
  const uint64_t* rc;        /* [ZKW_GL_RC_COUNT] round constants */
  const uint4* stream;
  const uint32_t* cursors;   /* [n_waves][4] */
  const uint32_t* dir;       /* [n_waves][max_cycles + 1][4] */
  const zkw_dev_scalars* scalars;
  const uint64_t* blob_digests; /* [n_blobs][4] */
  const uint2* blob_dir;
  const zkw_dev_preimage* preimages; /* [n_preimages] (midstate stage) */
  uint64_t* chunk_tails;     /* blob digests, first level: tail of every 64-word chunk (slot = first_word / 64 + blob + chunk) */
  uint64_t* midstates;       /* [n_preimages][12] sponge state after absorbing the code hash (decommit leaves) */
  uint32_t n_preimages;
  uint32_t reserved1;
  uint64_t* leaves;          /* [n_waves][cap][4] */
  uint32_t* idx;             /* [n_instances][per_instance_cap]; pooled: [n_waves][cap] */
  uint32_t* counts;          /* [n_instances] */
  uint32_t* offs;            /* pooled: [n_instances] first entry of the instance's list in idx[] */
  uint64_t* out;             /* chain: [n_instances][ZKW_QUEUE_COUNT][4]; blob chain: [n_blobs][4] */
} zkw_commit_params;

/* zkw_netstate_kernel: one instance per lane walks its cycles and nets its log queries against its frame events
 * (the device form of testing/storage.rs:144-186 + reference_impls/event_sink.rs:160-176) */
typedef struct zkw_netstate_params {
  uint32_t n_instances, L, n_waves, max_cycles, wave_threads;
  uint32_t cap_log, cap_aux;      /* stream capacity per wave (records) */
  uint32_t per_log, per_aux;      /* index-list capacity per instance */
  uint32_t hist_cap;              /* history capacity per instance (2 * per_log) */
  uint32_t mark_cap;              /* frame-mark capacity per instance */
  uint32_t storage_aux_byte, event_aux_byte, l1_aux_byte;
  uint32_t reserved0;
  const uint4* tails;             /* [n_waves][max_cycles][L] record tails: the per-cycle event counts (.w, low 24 bits) */
  const uint4* log_stream;
  const uint4* aux_stream;
  const zkw_dev_scalars* scalars;   /* status, n_cycles */
  const zkw_dev_scalars* scalars0;  /* pristine: the callstack depth the instance started with */
  const uint32_t* log_idx;  const uint32_t* log_cnt;   /* bucket pass over the log stream */
  const uint32_t* aux_idx;  const uint32_t* aux_cnt;   /* bucket pass over the aux stream, every type */
  uint32_t* st_hist;   /* [n_instances][hist_cap]  stream position | rollback << 31 */
  uint32_t* ev_hist;   /* [n_instances][hist_cap] */
  uint32_t* rb_st;     /* [n_instances][per_log]   pending storage rollbacks (stack) */
  uint32_t* rb_ev;     /* [n_instances][per_log]   pending event rollbacks = the surviving events at the end */
  uint32_t* marks;     /* [n_instances][mark_cap][2] */
  uint32_t* out_counts; /* [n_instances][4]: storage history, event history, surviving events, flags (1 = overflow) */
} zkw_netstate_params;

/* round constants: splitmix64 stream seeded with "zkwGLv1", values >= p rejected */
static inline void zkw_gl_round_constants(uint64_t* rc) {
  uint64_t x = 0x7a6b77474c7631ULL;
  int n = 0;
  while (n < ZKW_GL_RC_COUNT) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    if (z < 0xffffffff00000001ULL) rc[n++] = z;
  }
}
