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
#define ZKW_QUEUE_ID_BLOB 0xB10Bu
#define ZKW_COMMIT_STAGE_LEAF 0
#define ZKW_COMMIT_STAGE_BUCKET 1
#define ZKW_COMMIT_STAGE_CHAIN 2
#define ZKW_COMMIT_STAGE_BLOB_CHAIN 3

typedef struct zkw_commit_params {
  uint32_t n_instances, L, n_waves, max_cycles, wave_threads;
  uint32_t queue;            /* ZKW_QUEUE_* or ZKW_QUEUE_CODE_WORDS */
  uint32_t cap;              /* stream capacity per wave (records) */
  uint32_t per_instance_cap; /* idx capacity per instance */
  uint32_t n_blobs;
  uint32_t n_override;       /* != 0: number of records instead of cursors[] (code-word leaves) */
  const uint64_t* rc;        /* [ZKW_GL_RC_COUNT] round constants */
  const uint4* stream;
  const uint32_t* cursors;   /* [n_waves][4] */
  const uint32_t* dir;       /* [n_waves][max_cycles + 1][4] */
  const zkw_dev_scalars* scalars;
  const uint64_t* blob_digests; /* [n_blobs][4] */
  const uint2* blob_dir;
  uint64_t* leaves;          /* [n_waves][cap][4] */
  uint32_t* idx;             /* [n_instances][per_instance_cap] */
  uint32_t* counts;          /* [n_instances] */
  uint64_t* out;             /* chain: [n_instances][ZKW_QUEUE_COUNT][4]; blob chain: [n_blobs][4] */
} zkw_commit_params;

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
