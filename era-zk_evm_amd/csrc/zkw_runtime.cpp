// Host runtime of libzkw.so: implements the C ABI of include/zkw.h on top of the HIP kernels.
//
//   staging (zkw_batch_set_*)  ->  zkw_batch_upload: pristine device images
//   zkw_batch_reset            ->  device-side copy pristine -> working state (async)
//   zkw_batch_run              ->  zkw_cycle_kernel, bracketed by HIP events on the run stream
//   zkw_batch_sync             ->  waits, downloads the small per-instance scalars
//   zkw_batch_get_instance_trace -> downloads the owning wave's streams on demand and
//                                 de-interleaves them into the per-instance view
// There is no CPU execution path: without a GPU zkw_ctx_create fails with ZKW_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "zkw_commit.h"
#include "zkw_device.h"
#include "zkw_pack.h"

extern "C" hipError_t zkw_launch_cycle_kernel(const zkw_launch_args* A, hipStream_t stream);
extern "C" uint32_t zkw_cycle_kernel_lds_bytes(uint32_t L, uint32_t waves_per_group);
extern "C" hipError_t zkw_launch_reset_kernel(const zkw_fused_table* T, hipStream_t stream);
extern "C" hipError_t zkw_launch_commit(const zkw_fused_table* T, int stage, hipStream_t stream);
extern "C" hipError_t zkw_launch_pack(const zkw_pack_args* A, uint32_t wave_threads, uint32_t blocks, hipStream_t stream);
extern "C" hipError_t zkw_launch_restage(const zkw_restage_params* R, uint32_t wave_threads, hipStream_t stream);
extern "C" hipError_t zkw_launch_expand(const zkw_kparams* const* kp, void* const* dst, const uint32_t* n_waves, uint32_t n, uint64_t stride_i, uint64_t stride_k, uint32_t first,
                                        uint32_t count, uint32_t L, uint32_t wave_threads, uint32_t max_cycles_run, uint32_t n_cus, hipStream_t stream);

static_assert(sizeof(zkw_callstack_entry) == 112, "abi");
static_assert(sizeof(zkw_vm_local_state) == 680, "abi");
static_assert(sizeof(zkw_cycle_record) == 512, "abi");
static_assert(sizeof(zkw_mem_query) == 48, "abi");
static_assert(sizeof(zkw_log_query) == 128, "abi");
static_assert(sizeof(zkw_aux_event) == 256, "abi");
static_assert(sizeof(zkw_dev_entry) == 128, "dev");
static_assert(sizeof(zkw_dev_scalars) == 128, "dev");
static_assert(sizeof(zkw_dev_storage_entry) == 96, "dev");
static_assert(sizeof(zkw_dev_journal_entry) == 48, "dev");
static_assert(sizeof(zkw_dev_preimage) == 48, "dev");

struct zkw_ctx {
  int device = 0;
  int n_cus = 256;
  int wave_width = 64;
  bool has_isa = false;
  zkw_isa_table isa;
  uint2* d_isa = nullptr;
  std::string last_error;
  // test hooks / ablations (zkw_ctx_set_option; the environment is read once, in zkw_ctx_create)
  uint32_t opt_debug_flags = 0, opt_reset_skip = 0, opt_waves_per_group = 0, opt_lanes_per_wave = 0, opt_pack_blocks = 0;
  bool opt_read_values = false;
  uint32_t opt_staging_buffers = 0;
  uint32_t opt_link_flags_off = 0;
  uint32_t opt_link_selfcheck = 1;  // ZKW_OPT_LINK_SELFCHECK
  bool link_checked = false;        // the self-check of the link format has run on this context
  bool opt_no_inline_decommit = false, opt_debug_sync = false, opt_no_graph = false;
  // digests of code blobs already hashed on this context, keyed by a 128-bit content hash + length: batches that
  // share bytecode (the usual case) skip the sequential blob chain (~0.1 s for a 2000-word blob) at upload
  std::map<std::array<uint64_t, 3>, std::array<uint64_t, 4>> blob_digest_cache;
  // staging of zkw_blake2s256 (host buffers in, host buffers out); grown on demand, freed with the context
  void* b2_data = nullptr; size_t b2_data_cap = 0;
  uint64_t* b2_off = nullptr; size_t b2_off_cap = 0;
  void* b2_out = nullptr; size_t b2_out_cap = 0;
};

static std::string g_create_error;

#define HIP_TRY(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
      return ZKW_ERR_DEVICE;                                                                    \
    }                                                                                           \
  } while (0)

struct StagedInstance {
  zkw_vm_local_state state;
  std::vector<zkw_callstack_entry> inner;
  std::vector<std::pair<uint32_t, uint32_t>> code_pages;  // page -> blob
  std::vector<zkw_u256> heap;
  std::vector<zkw_u256> bootloader_calldata;  // BOOTLOADER_CALLDATA_PAGE (host only: the VM cannot reach it, zkw.h)
  std::vector<zkw_storage_slot> storage;
  bool has_state = false;
};

// What a witness is rebuilt onto on the host: the initial states the step ran on and the code of the batch (page -> blob, the
// blobs' words: a Code query's value does not travel).  Immutable once built and shared by reference count: the batch holds the
// inputs of its current staging, a delivery ticket those of ITS step — a restage, a new upload or the destruction of the batch
// while a ticket is still being read neither changes nor frees what the ticket rebuilds from.
struct CodeInputs {
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> pages;  // [instance] page -> blob, in registration order
  std::vector<std::vector<zkw_u256>> blobs;
  std::vector<uint32_t> preimage_blob;                             // [preimage] -> blob
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> frames0; // [instance] (base page, code page) of the INNER callstack entries,
                                                                   // outermost first (the current one is in the state): what the
                                                                   // implied pages of the link format are resolved against
  uint32_t time_delta = 0;                                         // consts.time_delta_per_cycle of the table the batch ran under
};
struct BatchInputs {
  std::vector<zkw_vm_local_state> states;  // [n]
  std::shared_ptr<const CodeInputs> code;
  // the heap images the step ran on ([n][heap_words], zero-padded) — what the shadow memory of the rebuild starts from when the
  // values of memory reads did not travel (ZKW_PACK_NO_READ_VALUES); heaps_known = false: the host does not hold them (a restage
  // with heap images through a ring of one staging buffer) and every value has to travel
  const zkw_u256* heap_data = nullptr;       // [n][heap_words]
  std::shared_ptr<const void> heap_owner;    // what keeps heap_data alive and unchanged: a vector (upload) or the pinned staging
                                             // buffer the images were restaged from (StageMem) — no copy either way
  uint32_t heap_words = 0;
  bool heaps_known = false;
};

// One pinned staging buffer of zkw_batch_restage / zkw_batch_staging ([states | heap images], instance-major).  Reference-counted:
// the batch holds it, and so does every BatchInputs whose heap images live in it — the batch does not hand such a buffer out again
// (zkw_batch_staging takes another one of its ring), and the memory is freed when the last of them is gone.
struct StageMem {
  uint8_t* h = nullptr;
  size_t bytes = 0;
  ~StageMem() {
    if (h) (void)hipHostFree(h);
  }
};
struct StageBuf {
  std::shared_ptr<StageMem> mem;
  hipEvent_t ev = nullptr;  // behind the H2D copies of the last restage out of it
  bool busy = false;
};

struct WaveTrace {  // de-interleaved streams of one wave
  std::vector<std::vector<zkw_cycle_record>> records;
  std::vector<std::vector<zkw_mem_query>> mem;
  std::vector<std::vector<zkw_log_query>> log;
  std::vector<std::vector<zkw_aux_event>> aux;
  std::vector<std::vector<uint32_t>> mem_off, log_off, aux_off;
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0) count = 1;
    return hipMalloc((void**)&p, count * sizeof(T));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  size_t bytes() const { return n * sizeof(T); }
};

struct zkw_batch {
  zkw_ctx* ctx;
  uint32_t n;
  zkw_limits lim;
  uint32_t L, n_waves;
  bool auto_L = false;  // limits.lanes_per_wave == 0: the wave width is settled at upload, when the programs are known
  uint32_t cap_mem, cap_log, cap_aux, cap_delta;
  // staging
  std::vector<std::vector<zkw_u256>> blobs;
  std::vector<std::pair<zkw_u256, uint32_t>> preimages;
  std::vector<StagedInstance> staged;
  std::shared_ptr<const BatchInputs> inputs;  // of the last upload / restage (what traces are rebuilt onto)
  zkw_block_properties props;
  bool uploaded = false, ran = false, synced = false;
  uint32_t cycles_run = 0;  // wave cycles since reset
  uint32_t heap_image_words = 0;
  uint32_t max_initial_depth = 0;  // deepest initial callstack among the instances
  // device: pristine
  DevBuf<uint4> d_regs0;
  DevBuf<zkw_dev_scalars> d_scalars0;
  DevBuf<zkw_dev_entry> d_callstack0;
  DevBuf<zkw_dev_frame_meta> d_frames0;
  DevBuf<zkw_dev_storage_entry> d_storage0;
  DevBuf<uint4> d_heap0;  // [n_waves][heap_image_words][2][L]
  // device: working
  DevBuf<uint4> d_regs;
  DevBuf<zkw_dev_scalars> d_scalars;
  DevBuf<zkw_dev_entry> d_callstack;
  DevBuf<zkw_dev_frame_meta> d_frames;
  DevBuf<zkw_dev_storage_entry> d_storage;
  DevBuf<zkw_dev_journal_entry> d_journal;
  DevBuf<zkw_dev_history> d_history;
  DevBuf<uint4> d_stack_vals, d_heap, d_aux;
  DevBuf<uint8_t> d_stack_ptrs;
  DevBuf<uint4> d_blob_words;
  DevBuf<uint2> d_blob_dir;
  DevBuf<zkw_dev_preimage> d_preimages;
  DevBuf<uint32_t> d_pre_index;  // open addressing over the code hashes (zkw_kparams.pre_index)
  uint32_t pre_mask = 0;
  // device: outputs
  DevBuf<uint4> d_tails, d_deltas, d_mem, d_log, d_auxs;
  DevBuf<uint32_t> d_wave_cycles, d_heap_dirty, d_storage_dirty;
  bool full_reset_pending = true;  // the first reset after an upload copies the whole heap image
  DevBuf<uint32_t> d_dir, d_cursors, d_krow;
  DevBuf<uint64_t> d_commit, d_rc, d_blob_digests, d_leaves, d_midstates;
  DevBuf<uint32_t> d_idx, d_counts, d_dq_count;
  DevBuf<uint64_t> d_dq_prev;  // [n][4]
  int dq_mode = 0;  // decommit-queue commitment since the last reset: 0 undecided, 1 chained by the cycle kernel, 2 by the commitment kernels
  // hipGraph of one whole step (reset -> cycle kernel -> commitment kernels), replayed by zkw_batch_step
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  uint32_t graph_cycles = 0, graph_mask = 0;
  int graph_dq_mode = 0;  // dq_mode the captured reset + run leave behind (re-established by a replay)
  hipStream_t graph_stream = nullptr;
  bool graph_failed = false;
  static const int EV_RING = 64;
  std::vector<hipEvent_t> evs;  // EV_RING (start, stop) pairs, one per run since the last sync
  uint32_t pending_runs = 0;
  hipStream_t run_stream = nullptr;
  // host results
  std::vector<zkw_dev_scalars> h_scalars;
  std::vector<uint32_t> h_cursors;
  std::map<uint32_t, std::unique_ptr<WaveTrace>> wave_cache;
  float kernel_ms = 0;
  uint32_t timed_runs = 0;
  zkw_kparams kp;            // host copy of the parameter block
  DevBuf<zkw_kparams> d_kp;  // device copy the kernels read (constant address space)
  // final net states (zkw_batch_net_states)
  DevBuf<uint32_t> d_ns_log_idx, d_ns_log_cnt, d_ns_aux_idx, d_ns_aux_cnt, d_ns_st_hist, d_ns_ev_hist, d_ns_rb_st, d_ns_rb_ev, d_ns_marks, d_ns_counts;
  DevBuf<zkw_commit_params> d_ns_bucket_params;  // [2]: log stream, aux stream (every type)
  DevBuf<zkw_netstate_params> d_ns_params;       // [1]
  bool ns_done = false;
  uint32_t ns_cached_wave = 0xffffffffu;
  std::vector<zkw_log_query> ns_wave_log;
  std::vector<zkw_log_query> ns_st_hist, ns_ev_hist;
  std::vector<zkw_event_message> ns_events, ns_l1;
  std::vector<zkw_storage_slot> ns_final;
  std::vector<StageBuf> stage;   // pinned staging of zkw_batch_restage: a small ring (ZKW_OPT_STAGING_BUFFERS), `stage_cur` the one in use ...
  uint32_t stage_cur = 0;
  DevBuf<uint8_t> d_stage;       // ... and where its H2D copies land (zkw_restage_kernel brings them into the device layouts)
  size_t d_stage_bytes = 0;
  bool heaps_restaged = false;   // the staged heap vectors are older than the device's images
  uint4* h_pack = nullptr;       // pinned block of the on-demand pack (one wave at a time: zkw_batch_get_instance_trace)
  uint64_t h_pack_units = 0;
  DevBuf<uint32_t> d_pack_state;  // allocation cursor + overflow flag of the pack kernel
  DevBuf<zkw_reset_params> d_reset_params;    // [1]
  DevBuf<zkw_commit_params> d_commit_params;  // [ZKW_QUEUE_COUNT] + [1] for the blob digests of the upload
};

static uint32_t pow2_ceil(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

extern "C" {

int zkw_ctx_create(int device, zkw_ctx** out) {
  if (!out) return ZKW_ERR_INVALID;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_error = std::string("no HIP device available (") + (e == hipSuccess ? "device count 0" : hipGetErrorString(e)) +
                     "): libzkw has no CPU fallback";
    return ZKW_ERR_DEVICE;
  }
  if (device < 0 || device >= count) {
    g_create_error = "device index out of range";
    return ZKW_ERR_INVALID;
  }
  auto* c = new zkw_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    g_create_error = "hipSetDevice failed";
    delete c;
    return ZKW_ERR_DEVICE;
  }
  (void)hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device);
  (void)hipDeviceGetAttribute(&c->wave_width, hipDeviceAttributeWarpSize, device);
  if (c->n_cus <= 0) c->n_cus = 256;
  if (c->wave_width <= 0 || c->wave_width > ZKW_WAVE) c->wave_width = ZKW_WAVE;
  {  // the only place the library reads its environment: initial values of the options, said out loud
    static const struct { const char* name; uint32_t opt; } envs[] = {
        {"ZKW_DEBUG_FLAGS", ZKW_OPT_DEBUG_FLAGS}, {"ZKW_RESET_SKIP", ZKW_OPT_RESET_SKIP}, {"ZKW_NO_INLINE_DECOMMIT", ZKW_OPT_NO_INLINE_DECOMMIT},
{"ZKW_DEBUG_SYNC", ZKW_OPT_DEBUG_SYNC}, {"ZKW_NO_GRAPH", ZKW_OPT_NO_GRAPH},
        {"ZKW_WAVES_PER_GROUP", ZKW_OPT_WAVES_PER_GROUP}, {"ZKW_LANES_PER_WAVE", ZKW_OPT_LANES_PER_WAVE}, {"ZKW_PACK_BLOCKS", ZKW_OPT_PACK_BLOCKS}};
    for (const auto& e : envs)
      if (const char* v = getenv(e.name)) {
        const uint64_t val = strtoull(v, nullptr, 0);
        fprintf(stderr, "libzkw: option %s=%llu taken from the environment (test hook / ablation)\n", e.name, (unsigned long long)val);
        (void)zkw_ctx_set_option(c, e.opt, val);
      }
  }
  *out = c;
  return ZKW_OK;
}

int zkw_ctx_set_option(zkw_ctx* c, uint32_t option, uint64_t value) {
  if (!c) return ZKW_ERR_INVALID;
  switch (option) {
    case ZKW_OPT_DEBUG_FLAGS: c->opt_debug_flags = (uint32_t)value; break;
    case ZKW_OPT_RESET_SKIP: c->opt_reset_skip = (uint32_t)value; break;
    case ZKW_OPT_NO_INLINE_DECOMMIT: c->opt_no_inline_decommit = value != 0; break;
    case ZKW_OPT_DEBUG_SYNC: c->opt_debug_sync = value != 0; break;
    case ZKW_OPT_NO_GRAPH: c->opt_no_graph = value != 0; break;
    case ZKW_OPT_WAVES_PER_GROUP:
      if (value > ZKW_MAX_WAVES_PER_GROUP) {
        c->last_error = "ZKW_OPT_WAVES_PER_GROUP: 0 (chosen per launch) or 1 .. 8";
        return ZKW_ERR_INVALID;
      }
      c->opt_waves_per_group = (uint32_t)value;
      break;
    case ZKW_OPT_LANES_PER_WAVE: c->opt_lanes_per_wave = (uint32_t)value; break;
    case ZKW_OPT_STAGING_BUFFERS: c->opt_staging_buffers = (uint32_t)std::min<uint64_t>(value, 64); break;
    case ZKW_OPT_READ_VALUES: c->opt_read_values = value != 0; break;
    case ZKW_OPT_LINK_FLAGS_OFF: c->opt_link_flags_off = (uint32_t)value; break;
    case ZKW_OPT_LINK_SELFCHECK: c->opt_link_selfcheck = (uint32_t)value; c->link_checked = false; break;
    case ZKW_OPT_PACK_BLOCKS: c->opt_pack_blocks = (uint32_t)value; break;
    default: c->last_error = "unknown option"; return ZKW_ERR_INVALID;
  }
  return ZKW_OK;
}

void zkw_ctx_destroy(zkw_ctx* c) {
  if (!c) return;
  if (c->d_isa) (void)hipFree(c->d_isa);
  if (c->b2_data) (void)hipFree(c->b2_data);
  if (c->b2_off) (void)hipFree(c->b2_off);
  if (c->b2_out) (void)hipFree(c->b2_out);
  delete c;
}

// =================================================================================================
// BLAKE2s-256 of a batch of byte strings (zkw_blake2s.hip) — the reference's `blake2` re-export, src/lib.rs:21
// =================================================================================================
extern "C" hipError_t zkw_launch_blake2s(const void* d_data, uint64_t total_bytes, const uint64_t* d_offsets, uint32_t n, void* d_digests,
                                         uint32_t wave_threads, hipStream_t stream);

int zkw_blake2s256_device(zkw_ctx* c, const void* d_data, uint64_t total_bytes, const uint64_t* d_offsets, uint32_t n_messages, void* d_digests,
                          void* hip_stream) {
  if (!c) return ZKW_ERR_INVALID;
  if (n_messages == 0) return ZKW_OK;
  if (!d_offsets || !d_digests || (!d_data && total_bytes != 0)) {
    c->last_error = "zkw_blake2s256_device: null buffer";
    return ZKW_ERR_INVALID;
  }
  if (((uintptr_t)d_data & 3u) != 0 || ((uintptr_t)d_digests & 15u) != 0 || ((uintptr_t)d_offsets & 7u) != 0) {
    c->last_error = "zkw_blake2s256_device: data must be 4-byte, offsets 8-byte and digests 16-byte aligned";
    return ZKW_ERR_INVALID;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, zkw_launch_blake2s(d_data, total_bytes, d_offsets, n_messages, d_digests, (uint32_t)c->wave_width, (hipStream_t)hip_stream));
  return ZKW_OK;
}

int zkw_blake2s256(zkw_ctx* c, const uint8_t* data, const uint64_t* offsets, uint32_t n_messages, uint8_t* digests) {
  if (!c) return ZKW_ERR_INVALID;
  if (n_messages == 0) return ZKW_OK;
  if (!offsets || !digests) {
    c->last_error = "zkw_blake2s256: null buffer";
    return ZKW_ERR_INVALID;
  }
  for (uint32_t i = 0; i < n_messages; i++)
    if (offsets[i + 1] < offsets[i]) {
      c->last_error = "zkw_blake2s256: offsets must not decrease";
      return ZKW_ERR_INVALID;
    }
  const uint64_t total = offsets[n_messages];
  if (total != 0 && !data) {
    c->last_error = "zkw_blake2s256: null data";
    return ZKW_ERR_INVALID;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  auto grow = [&](void** p, size_t* cap, size_t bytes) -> hipError_t {
    if (*cap >= bytes && *p) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    const hipError_t e = hipMalloc(p, bytes < 256 ? 256 : bytes);
    if (e == hipSuccess) *cap = bytes < 256 ? 256 : bytes;
    return e;
  };
  HIP_TRY(c, grow(&c->b2_data, &c->b2_data_cap, (size_t)((total + 3u) & ~(uint64_t)3u)));
  HIP_TRY(c, grow((void**)&c->b2_off, &c->b2_off_cap, ((size_t)n_messages + 1) * 8));
  HIP_TRY(c, grow(&c->b2_out, &c->b2_out_cap, (size_t)n_messages * 32));
  if (total) HIP_TRY(c, hipMemcpy(c->b2_data, data, (size_t)total, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->b2_off, offsets, ((size_t)n_messages + 1) * 8, hipMemcpyHostToDevice));
  HIP_TRY(c, zkw_launch_blake2s(c->b2_data, total, c->b2_off, n_messages, c->b2_out, (uint32_t)c->wave_width, nullptr));
  HIP_TRY(c, hipMemcpy(digests, c->b2_out, (size_t)n_messages * 32, hipMemcpyDeviceToHost));
  return ZKW_OK;
}

const char* zkw_last_error(zkw_ctx* c) { return c ? c->last_error.c_str() : g_create_error.c_str(); }

int zkw_ctx_set_isa(zkw_ctx* c, const zkw_isa_table* t) {
  if (!c || !t) return ZKW_ERR_INVALID;
  if (t->consts.panic_variant_idx >= ZKW_ISA_TABLE_SIZE || t->consts.nop_variant_idx >= ZKW_ISA_TABLE_SIZE) {
    c->last_error = "ISA table: panic/nop variant index out of range";
    return ZKW_ERR_INVALID;
  }
  {
    // the two masked encodings are the bare nop / panic variant: r0 operands, zero immediates, and a condition field that
    // names Always under this table's condition_lut (cycle.rs:135-148, 212-217)
    const zkw_isa_consts& k = t->consts;
    auto bare = [&](uint64_t enc, uint32_t idx) {
      const uint32_t field = (uint32_t)(enc >> 13) & 7u;
      return (enc & (ZKW_ISA_TABLE_SIZE - 1)) == idx && (enc >> 16) == 0 && ((enc >> 11) & 3u) == 0 && ((k.condition_lut >> (8 * field)) & 0xffu) == 0xffu;
    };
    if (!bare(k.exception_revert_encoding, k.panic_variant_idx) || !bare(k.nop_encoding, k.nop_variant_idx)) {
      c->last_error = "ISA table: nop/exception_revert encodings must be the bare nop/panic variant (Always, r0 operands, zero immediates)";
      return ZKW_ERR_INVALID;
    }
    const uint32_t regs[] = {k.call_regs & 0xffu, (k.call_regs >> 8) & 0xffu, (k.call_regs >> 16) & 0xffu, k.ret_regs & 0xffu, (k.ret_regs >> 8) & 0xffu,
                             (k.ret_regs >> 16) & 0xffu, k.ret_regs >> 24};
    bool ok = ((k.call_ranges >> 8) & 0xffu) <= ZKW_REGISTERS_COUNT && (k.call_ranges >> 24) <= ZKW_REGISTERS_COUNT;
    for (uint32_t r : regs) ok = ok && r < ZKW_REGISTERS_COUNT;
    if (!ok) {
      c->last_error = "ISA table: a far_call / ret register convention names a register beyond r15";
      return ZKW_ERR_INVALID;
    }
    // every row of condition_lut is the truth table of one of the eight Conditions over (lt | eq << 1 | gt << 2)
    // (cycle.rs:193-209: Always, Gt, Lt, Eq, Ge, Le, Ne, GtOrLt) — any other row would be a predicate the reference cannot name
    static const uint8_t kTruth[8] = {0xff, 0xf0, 0xaa, 0xcc, 0xfc, 0xee, 0x33, 0xfa};
    for (uint32_t f = 0; f < 8; f++) {
      const uint8_t row = (uint8_t)(k.condition_lut >> (8 * f));
      bool known = false;
      for (uint8_t t8 : kTruth) known = known || row == t8;
      if (!known) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "ISA table: condition_lut row %u (0x%02x) names no Condition", f, (unsigned)row);
        c->last_error = buf;
        return ZKW_ERR_INVALID;
      }
    }
    {  // FarCallForwardPageType: three different ABI bytes (far_call.rs:255, ret.rs:59)
      const uint32_t f0 = k.forwarding_codes & 0xffu, f1 = (k.forwarding_codes >> 8) & 0xffu, f2 = (k.forwarding_codes >> 16) & 0xffu;
      if (f0 == f1 || f0 == f2 || f1 == f2) {
        c->last_error = "ISA table: forwarding_codes must be three distinct bytes (UseHeap, ForwardFatPointer, UseAuxHeap)";
        return ZKW_ERR_INVALID;
      }
    }
    if (k.max_offset_for_add_sub == 0) {  // ptr.rs:47: with 0 every ptr.add / ptr.sub would panic
      c->last_error = "ISA table: max_offset_for_add_sub is zero";
      return ZKW_ERR_INVALID;
    }
  }
  c->isa = *t;
  std::vector<uint2> packed(ZKW_ISA_TABLE_SIZE);
  for (int i = 0; i < ZKW_ISA_TABLE_SIZE; i++) {
    const zkw_isa_entry& e = t->entries[i];
    if (e.opcode > 15 || e.src0_mode > 5 || e.dst0_mode > 3 || e.flags > 3 || e.props > 63 || e.variant > 15) {
      c->last_error = "ISA table: entry " + std::to_string(i) + " out of range";
      return ZKW_ERR_INVALID;
    }
    packed[i].x = ZKW_ATTR_PACK(e.opcode, e.variant, e.src0_mode, e.dst0_mode, e.flags, e.props) | zkw_short_class(e.opcode, e.variant, e.src0_mode, e.dst0_mode, e.props);
    packed[i].y = e.price;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->d_isa) HIP_TRY(c, hipMalloc((void**)&c->d_isa, sizeof(uint2) * ZKW_ISA_TABLE_SIZE));
  HIP_TRY(c, hipMemcpy(c->d_isa, packed.data(), sizeof(uint2) * ZKW_ISA_TABLE_SIZE, hipMemcpyHostToDevice));
  c->has_isa = true;
  return ZKW_OK;
}

int zkw_batch_create(zkw_ctx* c, uint32_t n, const zkw_limits* limits, zkw_batch** out) {
  if (!c || !limits || !out || n == 0) return ZKW_ERR_INVALID;
  if (!c->has_isa) {
    c->last_error = "zkw_ctx_set_isa must be called before zkw_batch_create";
    return ZKW_ERR_INVALID;
  }
  zkw_limits lim = *limits;
  if (lim.max_cycles == 0 || lim.max_far_frames == 0 || lim.max_callstack_depth == 0 || lim.stack_words == 0 || lim.heap_words == 0 || lim.aux_heap_words == 0) {
    c->last_error = "zkw_limits: zero capacity";
    return ZKW_ERR_INVALID;
  }
  {  // the kernel addresses a word of a wave's arena row with a 32-bit element offset (16-byte elements)
    const uint64_t words = std::max<uint64_t>(lim.stack_words, std::max<uint64_t>(lim.heap_words, lim.aux_heap_words));
    if ((uint64_t)lim.max_far_frames * words * 2u * ZKW_WAVE >= (1ull << 32)) {
      c->last_error = "zkw_limits: max_far_frames x page words exceeds 2^25 (a wave's arena row must stay below 2^32 elements)";
      return ZKW_ERR_LIMIT;
    }
  }
  lim.storage_slots = pow2_ceil(std::max(lim.storage_slots, 4u));
  if (lim.storage_journal == 0) lim.storage_journal = 4;
  if (lim.max_mem_queries == 0) lim.max_mem_queries = 6 * lim.max_cycles;
  if (lim.max_log_queries == 0) lim.max_log_queries = lim.max_cycles / 2 + 16;
  if (lim.max_aux_events == 0) lim.max_aux_events = lim.max_cycles / 4 + 16;
  if (lim.max_reg_deltas == 0) lim.max_reg_deltas = 2 * lim.max_cycles + 32;
  auto* b = new zkw_batch();
  b->ctx = c;
  b->n = n;
  b->lim = lim;
  uint32_t L = lim.lanes_per_wave;
  if (c->opt_lanes_per_wave) L = c->opt_lanes_per_wave;
  if (L == 0) {
    b->auto_L = true;
    // Measured on MI355X (profiles/r01_lane_sweep.md): the kernel is bound by the per-wave latency of one VM
    // cycle, which does not depend on the number of active lanes, so waves are filled: thin waves make a lone
    // launch of a small batch no faster (shared tape) and multiply the waves of a fused launch (256 x 4096 in
    // 256 batches: 0.8 G cycles/s with 4-lane waves, 19.6 G with full ones).  A caller whose instances run
    // DIFFERENT programs can ask for thinner waves (fewer opcode groups per wave-cycle); zkw_batch_upload does that
    // by itself when it sees that the instances were given different code (below).
    L = pow2_ceil(std::min<uint32_t>(n, (uint32_t)c->wave_width));
  }
  L = pow2_ceil(L);
  if (L > (uint32_t)c->wave_width) L = (uint32_t)c->wave_width;
  b->L = L;
  b->n_waves = (n + L - 1) / L;
  b->cap_mem = lim.max_mem_queries * L;
  b->cap_log = lim.max_log_queries * L;
  b->cap_aux = lim.max_aux_events * L;
  b->cap_delta = lim.max_reg_deltas * L;
  b->staged.resize(n);
  std::memset(&b->props, 0, sizeof b->props);
  b->blobs.emplace_back();  // blob 0 = the all-zero page (UNMAPPED_PAGE)
  *out = b;
  return ZKW_OK;
}

void zkw_batch_destroy(zkw_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->ctx->device);
  b->d_regs0.release(); b->d_scalars0.release(); b->d_callstack0.release(); b->d_frames0.release(); b->d_storage0.release(); b->d_heap0.release();
  b->d_regs.release(); b->d_scalars.release(); b->d_callstack.release(); b->d_frames.release(); b->d_storage.release(); b->d_journal.release();
  b->d_history.release(); b->d_stack_vals.release(); b->d_heap.release(); b->d_aux.release(); b->d_stack_ptrs.release(); b->d_blob_words.release();
  b->d_blob_dir.release(); b->d_preimages.release(); b->d_pre_index.release(); b->d_tails.release(); b->d_deltas.release(); b->d_wave_cycles.release(); b->d_heap_dirty.release(); b->d_storage_dirty.release(); b->d_mem.release(); b->d_log.release(); b->d_auxs.release();
  b->d_dir.release(); b->d_cursors.release(); b->d_krow.release(); b->d_kp.release(); b->d_reset_params.release(); b->d_commit_params.release();
  b->d_ns_log_idx.release(); b->d_ns_log_cnt.release(); b->d_ns_aux_idx.release(); b->d_ns_aux_cnt.release(); b->d_ns_st_hist.release();
  b->d_ns_ev_hist.release(); b->d_ns_rb_st.release(); b->d_ns_rb_ev.release(); b->d_ns_marks.release(); b->d_ns_counts.release();
  b->d_ns_bucket_params.release(); b->d_ns_params.release(); b->d_commit.release(); b->d_rc.release(); b->d_blob_digests.release(); b->d_leaves.release(); b->d_midstates.release();
  b->d_idx.release(); b->d_counts.release(); b->d_dq_count.release(); b->d_dq_prev.release(); b->d_pack_state.release();
  if (b->h_pack) (void)hipHostFree(b->h_pack);
  for (StageBuf& sb : b->stage) {
    if (sb.busy && sb.ev) (void)hipEventSynchronize(sb.ev);
    if (sb.ev) (void)hipEventDestroy(sb.ev);
  }
  b->stage.clear();  // (a buffer a delivered step still reads heap images from lives on until that ticket is released)
  b->d_stage.release();
  if (b->graph_exec) (void)hipGraphExecDestroy(b->graph_exec);
  if (b->graph) (void)hipGraphDestroy(b->graph);
  for (hipEvent_t e : b->evs) (void)hipEventDestroy(e);
  delete b;
}

// Any change of the staged inputs (and every upload) invalidates the captured step graph: it holds the launch
// geometry and the dynamic-LDS size of the upload it was captured under by value.
static void invalidate_inputs(zkw_batch* b) {
  b->uploaded = false;
  if (b->graph_exec) { (void)hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
  if (b->graph) { (void)hipGraphDestroy(b->graph); b->graph = nullptr; }
  b->graph_failed = false;
}

int zkw_batch_add_code_blob(zkw_batch* b, const zkw_u256* words, uint32_t n_words, uint32_t* blob_id) {
  if (!b || !blob_id || (n_words && !words)) return ZKW_ERR_INVALID;
  if (n_words > (1u << 16)) {  // MAX_CODE_PAGE_SIZE_IN_WORDS (memory.rs:276)
    b->ctx->last_error = "code blob longer than 2^16 words";
    return ZKW_ERR_INVALID;
  }
  b->blobs.emplace_back(words, words + n_words);
  *blob_id = (uint32_t)b->blobs.size() - 1;
  invalidate_inputs(b);
  return ZKW_OK;
}

int zkw_batch_add_decommit_preimage(zkw_batch* b, const zkw_u256* hash, uint32_t blob_id) {
  if (!b || !hash || blob_id >= b->blobs.size()) return ZKW_ERR_INVALID;
  for (auto& p : b->preimages)
    if (std::memcmp(&p.first, hash, 32) == 0) {
      b->ctx->last_error = "duplicate code hash (decommitter.rs:25)";
      return ZKW_ERR_INVALID;
    }
  b->preimages.emplace_back(*hash, blob_id);
  invalidate_inputs(b);
  return ZKW_OK;
}

int zkw_batch_set_code_page(zkw_batch* b, uint32_t first, uint32_t count, uint32_t page, uint32_t blob_id) {
  if (!b || (uint64_t)first + count > b->n || blob_id >= b->blobs.size()) return ZKW_ERR_INVALID;
  for (uint32_t i = first; i < first + count; i++) b->staged[i].code_pages.emplace_back(page, blob_id);
  invalidate_inputs(b);
  return ZKW_OK;
}

int zkw_batch_set_state(zkw_batch* b, uint32_t first, uint32_t count, const zkw_vm_local_state* states, const zkw_callstack_entry* inner,
                        uint32_t inner_depth) {
  if (!b || !states || (uint64_t)first + count > b->n || (inner_depth && !inner)) return ZKW_ERR_INVALID;
  if (inner_depth > b->lim.max_callstack_depth) {
    b->ctx->last_error = "initial callstack deeper than limits.max_callstack_depth";
    return ZKW_ERR_LIMIT;
  }
  for (uint32_t i = 0; i < count; i++) {
    StagedInstance& s = b->staged[first + i];
    if (states[i].callstack_depth != inner_depth) {
      b->ctx->last_error = "state.callstack_depth != inner_depth";
      return ZKW_ERR_INVALID;
    }
    s.state = states[i];
    s.inner.assign(inner + (size_t)i * inner_depth, inner + (size_t)(i + 1) * inner_depth);
    s.has_state = true;
  }
  invalidate_inputs(b);
  return ZKW_OK;
}

int zkw_batch_set_heap(zkw_batch* b, uint32_t instance, const zkw_u256* words, uint32_t n_words) {
  if (!b || instance >= b->n || (n_words && !words)) return ZKW_ERR_INVALID;
  if (n_words > b->lim.heap_words) {
    b->ctx->last_error = "heap image longer than limits.heap_words";
    return ZKW_ERR_LIMIT;
  }
  b->staged[instance].heap.assign(words, words + n_words);
  b->heaps_restaged = false;  // (the caller is setting heaps again: a re-upload uses these)
  invalidate_inputs(b);
  return ZKW_OK;
}

int zkw_batch_set_bootloader_calldata(zkw_batch* b, uint32_t instance, const zkw_u256* words, uint32_t n_words) {
  if (!b || instance >= b->n || (n_words && !words)) return ZKW_ERR_INVALID;
  b->staged[instance].bootloader_calldata.assign(words, words + n_words);  // nothing on the device depends on it
  return ZKW_OK;
}

int zkw_batch_set_storage(zkw_batch* b, uint32_t instance, const zkw_storage_slot* slots, uint32_t n_slots) {
  if (!b || instance >= b->n || (n_slots && !slots)) return ZKW_ERR_INVALID;
  if (n_slots * 2 > b->lim.storage_slots) {
    b->ctx->last_error = "storage snapshot needs limits.storage_slots >= 2 * n_slots";
    return ZKW_ERR_LIMIT;
  }
  b->staged[instance].storage.assign(slots, slots + n_slots);
  invalidate_inputs(b);
  return ZKW_OK;
}

int zkw_batch_set_block_properties(zkw_batch* b, const zkw_block_properties* p) {
  if (!b || !p) return ZKW_ERR_INVALID;
  b->props = *p;
  if (b->uploaded) {  // keep the device parameter block current (takes effect for runs enqueued after this call)
    zkw_ctx* c = b->ctx;
    b->kp.props = *p;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(&b->d_kp.p->props, p, sizeof *p, hipMemcpyHostToDevice));
  }
  return ZKW_OK;
}

// same hash / probe sequence as storage_find() in zkw_kernels.hip
static uint32_t host_storage_hash(uint32_t shard, const uint32_t addr[5], const uint32_t key[8]) {
  uint32_t h = 0x9e3779b9u * (shard + 1);
  for (int i = 0; i < 8; i++) h = (h ^ key[i]) * 0x85ebca6bu, h ^= h >> 15;
  for (int i = 0; i < 5; i++) h = (h ^ addr[i]) * 0xc2b2ae35u, h ^= h >> 13;
  return h;
}

static void entry_to_dev(const zkw_callstack_entry& e, uint32_t blob, uint32_t slot, zkw_dev_entry* o) {
  std::memset(o, 0, sizeof *o);
  o->e = e;
  o->e.reserved0 = 0;
  o->e.reserved1 = 0;
  o->code_blob = blob;
  o->frame_slot = slot;
  o->journal_mark = 0;
}

// The device images of instance i — register file, scalars, callstack rows, frame metas, heap image — written at the
// instance's places of the batch-wide arrays (zkw_batch_upload formats every instance, zkw_batch_restage the ones it is given;
// instances write disjoint elements, so host threads may format different instances at once).  `frames` rows must be zeroed.
static int format_instance(const zkw_batch* b, uint32_t i, uint32_t himg, uint4* regs, zkw_dev_scalars* scal, zkw_dev_entry* stack, zkw_dev_frame_meta* frames, uint4* heap0,
                           std::string* err) {
  const StagedInstance& s = b->staged[i];
  const uint32_t L = b->L, F = b->lim.max_far_frames, D = b->lim.max_callstack_depth;
  const uint32_t w = i / L, l = i % L;
  const zkw_vm_local_state& st = s.state;
  for (int r = 0; r < ZKW_REGISTERS_COUNT; r++) {
    const uint32_t* v = (const uint32_t*)st.registers[r].l;
    regs[((size_t)w * ZKW_REG_CHUNKS + 2 * r) * L + l] = make_uint4(v[0], v[1], v[2], v[3]);
    regs[((size_t)w * ZKW_REG_CHUNKS + 2 * r + 1) * L + l] = make_uint4(v[4], v[5], v[6], v[7]);
  }
  zkw_dev_scalars& sc = scal[i];
  std::memcpy(sc.prev_code_word, st.previous_code_word.l, 32);
  std::memcpy(sc.ctx_u128_reg, st.context_u128_register, 16);
  sc.ptr_bitmap = st.register_ptr_bitmap & 0x7fffu;
  sc.flags = (st.flags & 7u) | (st.pending_exception ? 8u : 0u);
  sc.prev_code_page = st.previous_code_memory_page;
  sc.timestamp = st.timestamp;
  sc.cycle_counter = st.monotonic_cycle_counter;
  sc.spent_pubdata = st.spent_pubdata_counter;
  sc.memory_page_counter = st.memory_page_counter;
  sc.absolute_execution_step = st.absolute_execution_step;
  sc.ergs_per_pubdata = st.current_ergs_per_pubdata_byte;
  sc.tx_number = st.tx_number_in_block;
  sc.prev_super_pc = st.previous_super_pc;
  sc.depth = st.callstack_depth;
  sc.status = ZKW_STATUS_RUNNING;
  sc.n_cycles = 0;
  sc.first_dynamic_page = st.memory_page_counter;
  // callstack: entries 0..depth-1 = inner, entry depth = current.  Far frames alive at reset get
  // arena slots in order (push_bootloader_context's start_global_frame, helpers.rs:306-315).
  uint32_t next_slot = 0;
  auto blob_of = [&](uint32_t page) -> int64_t {
    if (page == 0) return 0;
    for (auto it = s.code_pages.rbegin(); it != s.code_pages.rend(); ++it)
      if (it->first == page) return it->second;
    return -1;
  };
  const uint32_t depth = st.callstack_depth;
  for (uint32_t d = 0; d <= depth; d++) {
    const zkw_callstack_entry& e = d < depth ? s.inner[d] : st.current;
    uint32_t slot = 0;
    if (d > 0) {
      if (!e.is_local_frame) {
        if (next_slot >= F) {
          *err = "initial far frames exceed limits.max_far_frames";
          return ZKW_ERR_LIMIT;
        }
        frames[(size_t)i * F + next_slot].base_page = e.base_memory_page;
        slot = next_slot++;
      } else {
        slot = next_slot ? next_slot - 1 : 0;
      }
    }
    int64_t blob = blob_of(e.code_page);
    if (blob < 0) {
      *err = "instance " + std::to_string(i) + ": code page " + std::to_string(e.code_page) + " has no blob (zkw_batch_set_code_page)";
      return ZKW_ERR_INVALID;
    }
    entry_to_dev(e, (uint32_t)blob, slot, &stack[(size_t)i * (D + 1) + d]);
  }
  if (next_slot == 0) next_slot = 1;  // slot 0 is reserved even for an already-ended VM
  sc.n_initial_slots = next_slot;
  sc.next_slot = next_slot;
  for (uint32_t k = next_slot; k < F; k++) frames[(size_t)i * F + k].stack_hwm = 0xffffffffu;  // ZKW_SLOT_FREE (zkw_kernels.hip)
  // heap image of the current frame
  if (!s.heap.empty()) {
    const uint32_t cur_slot = stack[(size_t)i * (D + 1) + depth].frame_slot;
    if (cur_slot != 0) {
      *err = "zkw_batch_set_heap supports the first far frame only";
      return ZKW_ERR_INVALID;
    }
    frames[(size_t)i * F + cur_slot].heap_hwm = (uint32_t)s.heap.size();
    for (size_t k = 0; k < s.heap.size(); k++) {
      const uint32_t* v = (const uint32_t*)s.heap[k].l;
      heap0[(((size_t)w * himg + k) * 2) * L + l] = make_uint4(v[0], v[1], v[2], v[3]);      // [word][2][L]: low halves,
      heap0[(((size_t)w * himg + k) * 2 + 1) * L + l] = make_uint4(v[4], v[5], v[6], v[7]);  // then high halves
    }
  }
  return ZKW_OK;
}

int zkw_batch_upload(zkw_batch* b) {
  if (!b) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  invalidate_inputs(b);  // a re-upload may change the geometry (wave width, wave count) a captured step graph holds by value
  if (b->heaps_restaged) {
    c->last_error = "zkw_batch_upload: the heap images were replaced by zkw_batch_restage and the staged ones are stale: zkw_batch_set_heap again first";
    return ZKW_ERR_INVALID;
  }
  for (uint32_t i = 0; i < b->n; i++)
    if (!b->staged[i].has_state) {
      c->last_error = "instance " + std::to_string(i) + " has no state (zkw_batch_set_state)";
      return ZKW_ERR_INVALID;
    }
  if (b->auto_L) {
    // Wave width when the caller left it to the library.  Lanes of a wave that hold different instruction words are
    // served one after the other (opcode-word grouping), so instances that were given DIFFERENT code get thin waves:
    // as few lanes per wave as still fill the chip's wave slots once (8 per CU: two waves per SIMD).  Measured with 4096 different
    // arithmetic tapes (profiles/tools/divergent_tapes.py): 28 M cycles/s with 64-lane waves, 284 M with 4-lane waves.
    // Instances that share their code keep full waves.
    bool same_code = true;
    for (uint32_t i = 1; i < b->n && same_code; i++) same_code = b->staged[i].code_pages == b->staged[0].code_pages;
    uint32_t L2 = pow2_ceil(std::min<uint32_t>(b->n, (uint32_t)c->wave_width));
    if (!same_code) {
      const uint32_t slots = 8u * (uint32_t)std::max(1, c->n_cus);
      L2 = std::min<uint32_t>(L2, pow2_ceil((b->n + slots - 1) / slots));
    }
    b->L = L2;
    b->n_waves = (b->n + L2 - 1) / L2;
    b->cap_mem = b->lim.max_mem_queries * L2;
    b->cap_log = b->lim.max_log_queries * L2;
    b->cap_aux = b->lim.max_aux_events * L2;
    b->cap_delta = b->lim.max_reg_deltas * L2;
  }
  const uint32_t n = b->n, L = b->L, W = b->n_waves;
  const zkw_limits& lim = b->lim;
  const uint32_t F = lim.max_far_frames, D = lim.max_callstack_depth;
  // ---- blobs ----
  std::vector<uint2> dir(b->blobs.size());
  size_t total_words = 0;
  for (size_t i = 0; i < b->blobs.size(); i++) {
    dir[i].x = (uint32_t)total_words;
    dir[i].y = (uint32_t)b->blobs[i].size();
    total_words += b->blobs[i].size();
  }
  std::vector<zkw_u256> all(total_words ? total_words : 1);
  for (size_t i = 0; i < b->blobs.size(); i++)
    if (!b->blobs[i].empty()) std::memcpy(&all[dir[i].x], b->blobs[i].data(), b->blobs[i].size() * 32);
  b->d_blob_words.release();
  b->d_blob_dir.release();
  b->d_preimages.release();
  HIP_TRY(c, b->d_blob_words.alloc(all.size() * 2));
  HIP_TRY(c, hipMemcpy(b->d_blob_words.p, all.data(), all.size() * 32, hipMemcpyHostToDevice));
  HIP_TRY(c, b->d_blob_dir.alloc(dir.size()));
  HIP_TRY(c, hipMemcpy(b->d_blob_dir.p, dir.data(), dir.size() * sizeof(uint2), hipMemcpyHostToDevice));
  std::vector<zkw_dev_preimage> pre(b->preimages.size() ? b->preimages.size() : 1);
  std::memset(pre.data(), 0, pre.size() * sizeof(zkw_dev_preimage));
  for (size_t i = 0; i < b->preimages.size(); i++) {
    std::memcpy(pre[i].hash, &b->preimages[i].first, 32);
    pre[i].blob = b->preimages[i].second;
  }
  HIP_TRY(c, b->d_preimages.alloc(pre.size()));
  HIP_TRY(c, hipMemcpy(b->d_preimages.p, pre.data(), pre.size() * sizeof(zkw_dev_preimage), hipMemcpyHostToDevice));
  {  // known_hashes as a lookup structure (decommitter.rs:23-28, 52): open addressing over the code hashes, at most half full
    const uint32_t slots = pow2_ceil(std::max<uint32_t>(4u, 2u * (uint32_t)b->preimages.size()));
    std::vector<uint32_t> index(slots, 0u);
    for (size_t i = 0; i < b->preimages.size(); i++) {
      uint32_t at = zkw_pre_hash(pre[i].hash) & (slots - 1);
      while (index[at]) at = (at + 1) & (slots - 1);
      index[at] = (uint32_t)i + 1;
    }
    b->pre_mask = slots - 1;
    b->d_pre_index.release();
    HIP_TRY(c, b->d_pre_index.alloc(slots));
    HIP_TRY(c, hipMemcpy(b->d_pre_index.p, index.data(), (size_t)slots * 4, hipMemcpyHostToDevice));
  }

  // ---- blob digests for the decommit-queue commitment, once per upload ----
  {
    uint64_t rc[ZKW_GL_RC_COUNT];
    zkw_gl_round_constants(rc);
    b->d_rc.release();
    HIP_TRY(c, b->d_rc.alloc(ZKW_GL_RC_COUNT));
    HIP_TRY(c, hipMemcpy(b->d_rc.p, rc, sizeof rc, hipMemcpyHostToDevice));
    b->d_blob_digests.release();
    HIP_TRY(c, b->d_blob_digests.alloc(b->blobs.size() * 4));
    DevBuf<uint64_t> word_leaves;
    HIP_TRY(c, word_leaves.alloc(all.size() * 4));
    zkw_commit_params C;
    std::memset(&C, 0, sizeof C);
    C.n_waves = 1; C.wave_threads = (uint32_t)c->wave_width; C.queue = ZKW_QUEUE_CODE_WORDS; C.cap = (uint32_t)total_words;
    C.n_override = (uint32_t)total_words; C.n_blobs = (uint32_t)b->blobs.size(); C.rc = b->d_rc.p; C.stream = b->d_blob_words.p;
    C.blob_dir = b->d_blob_dir.p; C.leaves = word_leaves.p; C.out = b->d_blob_digests.p;
    if (!b->d_commit_params.p) HIP_TRY(c, b->d_commit_params.alloc(ZKW_QUEUE_COUNT + 1));
    HIP_TRY(c, hipMemcpy(b->d_commit_params.p + ZKW_QUEUE_COUNT, &C, sizeof C, hipMemcpyHostToDevice));
    zkw_fused_table T;
    std::memset(&T, 0, sizeof T);
    T.reserved[0] = ZKW_QUEUE_CODE_WORDS;
    T.p[0] = b->d_commit_params.p + ZKW_QUEUE_COUNT; T.n = 1; T.max_waves = 1; T.max_cap = C.cap; T.wave_threads = C.wave_threads; T.n_blobs = C.n_blobs;
    std::vector<std::array<uint64_t, 3>> keys(b->blobs.size());
    std::vector<uint64_t> digests(b->blobs.size() * 4);
    bool all_cached = true;
    for (size_t i = 0; i < b->blobs.size(); i++) {
      uint64_t h1 = 0xcbf29ce484222325ULL, h2 = 0x9e3779b97f4a7c15ULL;
      for (const zkw_u256& w : b->blobs[i])
        for (int k = 0; k < 4; k++) {
          h1 = (h1 ^ w.l[k]) * 0x100000001b3ULL;
          h2 = (h2 + w.l[k]) * 0xff51afd7ed558ccdULL;
          h2 ^= h2 >> 29;
        }
      keys[i] = {h1, h2, (uint64_t)b->blobs[i].size()};
      auto it = c->blob_digest_cache.find(keys[i]);
      if (it == c->blob_digest_cache.end()) all_cached = false;
      else std::memcpy(&digests[4 * i], it->second.data(), 32);
    }
    if (all_cached) {
      if (!digests.empty()) HIP_TRY(c, hipMemcpy(b->d_blob_digests.p, digests.data(), digests.size() * 8, hipMemcpyHostToDevice));
    } else {
      DevBuf<uint64_t> chunk_tails;
      HIP_TRY(c, chunk_tails.alloc((total_words / ZKW_BLOB_CHUNK_WORDS + b->blobs.size() + 2) * 4));
      C.chunk_tails = chunk_tails.p;
      HIP_TRY(c, hipMemcpy(b->d_commit_params.p + ZKW_QUEUE_COUNT, &C, sizeof C, hipMemcpyHostToDevice));
      if (total_words) HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_LEAF, nullptr));
      if (total_words) HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_BLOB_CHUNKS, nullptr));
      HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_BLOB_CHAIN, nullptr));
      HIP_TRY(c, hipStreamSynchronize(nullptr));
      chunk_tails.release();
      HIP_TRY(c, hipStreamSynchronize(nullptr));
      if (!digests.empty()) HIP_TRY(c, hipMemcpy(digests.data(), b->d_blob_digests.p, digests.size() * 8, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < b->blobs.size(); i++) {
        std::array<uint64_t, 4> d;
        std::memcpy(d.data(), &digests[4 * i], 32);
        c->blob_digest_cache[keys[i]] = d;
      }
    }
    word_leaves.release();
    // sponge midstates of the decommit leaves, one per (hash -> blob) pair
    b->d_midstates.release();
    HIP_TRY(c, b->d_midstates.alloc(std::max<size_t>(1, b->preimages.size()) * 12));
    if (!b->preimages.empty()) {
      C.preimages = b->d_preimages.p; C.midstates = b->d_midstates.p; C.n_preimages = (uint32_t)b->preimages.size();
      C.blob_digests = b->d_blob_digests.p;  // the leaf of a decommit commits to the digest of the blob the hash maps to
      HIP_TRY(c, hipMemcpy(b->d_commit_params.p + ZKW_QUEUE_COUNT, &C, sizeof C, hipMemcpyHostToDevice));
      T.n_blobs = C.n_preimages;
      HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_MIDSTATE, nullptr));
      HIP_TRY(c, hipStreamSynchronize(nullptr));
    }
  }

  // ---- per-instance pristine images ----
  std::vector<uint4> regs((size_t)W * ZKW_REG_CHUNKS * L, make_uint4(0, 0, 0, 0));
  std::vector<zkw_dev_scalars> scal(n);
  std::vector<zkw_dev_entry> stack((size_t)n * (D + 1));
  std::vector<zkw_dev_frame_meta> frames((size_t)n * F);
  std::vector<zkw_dev_storage_entry> storage((size_t)n * lim.storage_slots);
  std::memset(scal.data(), 0, scal.size() * sizeof(zkw_dev_scalars));
  std::memset(stack.data(), 0, stack.size() * sizeof(zkw_dev_entry));
  std::memset(frames.data(), 0, frames.size() * sizeof(zkw_dev_frame_meta));
  std::memset(storage.data(), 0, storage.size() * sizeof(zkw_dev_storage_entry));
  uint32_t himg = 0;
  for (uint32_t i = 0; i < n; i++) himg = std::max(himg, (uint32_t)b->staged[i].heap.size());
  b->heap_image_words = himg;
  b->max_initial_depth = 0;
  std::vector<uint4> heap0((size_t)W * himg * L * 2, make_uint4(0, 0, 0, 0));
  for (uint32_t i = 0; i < n; i++) {
    const StagedInstance& s = b->staged[i];
    b->max_initial_depth = std::max(b->max_initial_depth, s.state.callstack_depth);
    {
      std::string err;
      const int frc = format_instance(b, i, himg, regs.data(), scal.data(), stack.data(), frames.data(), heap0.data(), &err);
      if (frc != ZKW_OK) {
        c->last_error = err;
        return frc;
      }
    }
    // storage snapshot
    zkw_dev_storage_entry* tab = &storage[(size_t)i * lim.storage_slots];
    const uint32_t mask = lim.storage_slots - 1;
    for (const zkw_storage_slot& sl : s.storage) {
      uint32_t key[8], addr[5];
      std::memcpy(key, sl.key.l, 32);
      std::memcpy(addr, sl.address, 20);
      uint32_t pos = host_storage_hash(sl.shard_id, addr, key) & mask;
      for (;;) {
        zkw_dev_storage_entry& e = tab[pos];
        bool same = (e.shard_state & 0x100u) && (e.shard_state & 0xffu) == sl.shard_id && !std::memcmp(e.key, key, 32) && !std::memcmp(e.address, addr, 20);
        if (!(e.shard_state & 0x100u) || same) {
          std::memcpy(e.key, key, 32);
          std::memcpy(e.address, addr, 20);
          std::memcpy(e.value, sl.value.l, 32);
          e.shard_state = sl.shard_id | 0x100u | 0x400u;  // occupied + present in the reference's `inner` map
          break;
        }
        pos = (pos + 1) & mask;
      }
    }
  }
  auto up = [&](auto& dbuf, const auto& host) -> hipError_t {
    dbuf.release();
    hipError_t e = dbuf.alloc(host.size());
    if (e != hipSuccess) return e;
    if (host.empty()) return hipSuccess;
    return hipMemcpy(dbuf.p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice);
  };
  HIP_TRY(c, up(b->d_regs0, regs));
  HIP_TRY(c, up(b->d_scalars0, scal));
  HIP_TRY(c, up(b->d_callstack0, stack));
  HIP_TRY(c, up(b->d_frames0, frames));
  HIP_TRY(c, up(b->d_storage0, storage));
  HIP_TRY(c, up(b->d_heap0, heap0));
  // ---- working state + arenas + outputs ----
  auto ensure = [&](auto& dbuf, size_t count) -> hipError_t {
    if (dbuf.p && dbuf.n == count) return hipSuccess;
    dbuf.release();
    return dbuf.alloc(count);
  };
  HIP_TRY(c, ensure(b->d_regs, regs.size()));
  HIP_TRY(c, ensure(b->d_scalars, scal.size()));
  HIP_TRY(c, ensure(b->d_callstack, stack.size()));
  HIP_TRY(c, ensure(b->d_frames, frames.size()));
  HIP_TRY(c, ensure(b->d_storage, storage.size()));
  HIP_TRY(c, ensure(b->d_journal, (size_t)n * lim.storage_journal));
  const uint32_t hist_pitch = std::max<uint32_t>(1u, (uint32_t)b->preimages.size());  // one row per known code hash: 8 bytes, half a 16-byte unit — pitch rounded to even
  const uint32_t hist_pitch2 = (hist_pitch + 1u) & ~1u;
  HIP_TRY(c, ensure(b->d_history, (size_t)n * hist_pitch2));
  HIP_TRY(c, ensure(b->d_stack_vals, (size_t)W * F * lim.stack_words * L * 2));
  HIP_TRY(c, ensure(b->d_stack_ptrs, (size_t)W * F * lim.stack_words * L));
  HIP_TRY(c, ensure(b->d_heap, (size_t)W * F * lim.heap_words * L * 2));
  HIP_TRY(c, ensure(b->d_aux, (size_t)W * F * lim.aux_heap_words * L * 2));
  HIP_TRY(c, ensure(b->d_tails, (size_t)W * lim.max_cycles * L));
  HIP_TRY(c, ensure(b->d_deltas, (size_t)W * b->cap_delta * 2));
  HIP_TRY(c, ensure(b->d_wave_cycles, (size_t)W));
  HIP_TRY(c, ensure(b->d_heap_dirty, std::max<size_t>(1, (size_t)W * ((b->heap_image_words + 31) / 32) * L)));
  HIP_TRY(c, ensure(b->d_storage_dirty, std::max<size_t>(1, (size_t)b->n * ((b->lim.storage_slots + 31) / 32))));
  b->full_reset_pending = true;
  HIP_TRY(c, ensure(b->d_mem, (size_t)W * b->cap_mem * 3));
  HIP_TRY(c, ensure(b->d_log, (size_t)W * b->cap_log * 8));
  HIP_TRY(c, ensure(b->d_auxs, (size_t)W * b->cap_aux * 16));
  HIP_TRY(c, ensure(b->d_dir, (size_t)W * (lim.max_cycles + 1) * 4));
  HIP_TRY(c, ensure(b->d_cursors, (size_t)W * 4));
  HIP_TRY(c, ensure(b->d_krow, (size_t)W * ZKW_KROW_WORDS * L));
  HIP_TRY(c, ensure(b->d_commit, (size_t)n * ZKW_QUEUE_COUNT * 4));
  HIP_TRY(c, ensure(b->d_dq_count, (size_t)n));
  HIP_TRY(c, ensure(b->d_dq_prev, (size_t)n * 4));
  if (b->evs.empty()) {
    b->evs.resize(2 * zkw_batch::EV_RING, nullptr);
    for (auto& e : b->evs) HIP_TRY(c, hipEventCreate(&e));
  }
  // kernel parameter block
  zkw_kparams& P = b->kp;
  std::memset(&P, 0, sizeof P);
  P.n_instances = n; P.L = L; P.n_waves = W; P.max_cycles = lim.max_cycles;
  P.F = F; P.D = D; P.S = lim.stack_words; P.H = lim.heap_words; P.A = lim.aux_heap_words;
  P.storage_slots = lim.storage_slots; P.storage_journal = lim.storage_journal;
  P.cap_mem = b->cap_mem; P.cap_log = b->cap_log; P.cap_aux = b->cap_aux;
  P.n_blobs = (uint32_t)b->blobs.size(); P.n_preimages = (uint32_t)b->preimages.size();
  P.consts = c->isa.consts;
  P.wave_threads = (uint32_t)c->wave_width;
  P.waves_per_group = 0;  // (chosen per launch: zkw_launch_args.waves_per_group, pick_waves_per_group)
  P.isa = c->d_isa;
  P.krow = b->d_krow.p;
  P.regs = b->d_regs.p; P.scalars = b->d_scalars.p; P.callstack = b->d_callstack.p; P.frames = b->d_frames.p;
  P.stack_vals = b->d_stack_vals.p; P.stack_ptrs = b->d_stack_ptrs.p; P.heap = b->d_heap.p; P.aux_heap = b->d_aux.p;
  P.storage = b->d_storage.p; P.journal = b->d_journal.p; P.history = b->d_history.p; P.hist_pitch = hist_pitch2;
  P.pre_index = b->d_pre_index.p; P.pre_mask = b->pre_mask;
  P.blob_words = b->d_blob_words.p; P.blob_dir = b->d_blob_dir.p; P.preimages = b->d_preimages.p;
  P.commit_rc = b->d_rc.p; P.midstates = b->d_midstates.p; P.blob_digests = b->d_blob_digests.p; P.commit_out = b->d_commit.p; P.dq_count = b->d_dq_count.p; P.dq_prev = b->d_dq_prev.p;
  P.tails = b->d_tails.p; P.deltas = b->d_deltas.p; P.wave_cycles = b->d_wave_cycles.p; P.cap_delta = b->cap_delta; P.heap_dirty = b->d_heap_dirty.p; P.heap_image_words = b->heap_image_words; P.mem_stream = b->d_mem.p; P.log_stream = b->d_log.p; P.aux_stream = b->d_auxs.p;
  P.dir = b->d_dir.p; P.cursors = b->d_cursors.p; P.callstack0 = b->d_callstack0.p;
  P.regs0 = b->d_regs0.p; P.scalars0 = b->d_scalars0.p; P.storage_dirty = b->d_storage_dirty.p;
  P.props = b->props;
  HIP_TRY(c, ensure(b->d_kp, 1));
  HIP_TRY(c, hipMemcpy(b->d_kp.p, &P, sizeof P, hipMemcpyHostToDevice));
  {  // parameter blocks of the reset and commitment kernels (device copies, constant per upload)
    zkw_reset_params R;
    std::memset(&R, 0, sizeof R);
    // register files and scalars are not restored: a wave's first launch after a reset reads the pristine images
    // (zkw_kparams.regs0 / scalars0) and its write-back fills the working buffers, which nothing reads before a run
    R.dst[0] = b->d_regs.p; R.src[0] = b->d_regs0.p; R.n16[0] = 0;
    R.dst[1] = (uint4*)b->d_scalars.p; R.src[1] = (const uint4*)b->d_scalars0.p; R.n16[1] = 0;
    R.dst[2] = (uint4*)b->d_callstack.p; R.src[2] = (const uint4*)b->d_callstack0.p; R.cs_pitch16 = (b->lim.max_callstack_depth + 1) * (uint32_t)(sizeof(zkw_dev_entry) / 16);
    R.cs_row16 = std::min(R.cs_pitch16, (b->max_initial_depth + 1) * (uint32_t)(sizeof(zkw_dev_entry) / 16));
    R.n16[2] = b->n * R.cs_row16;
    R.dst[3] = (uint4*)b->d_frames.p; R.src[3] = (const uint4*)b->d_frames0.p; R.n16[3] = (uint32_t)(b->d_frames0.bytes() / 16);
    R.dst[4] = (uint4*)b->d_storage.p; R.src[4] = (const uint4*)b->d_storage0.p; R.n16[4] = (uint32_t)(b->d_storage0.bytes() / 16);
    R.heap_dst = b->d_heap.p; R.heap_src = b->d_heap0.p;
    R.heap_row16 = b->heap_image_words * b->L * 2;
    R.heap_pitch16 = b->lim.max_far_frames * b->lim.heap_words * b->L * 2;
    R.n_waves = b->n_waves;
    R.cursors = b->d_cursors.p;
    R.wave_cycles = b->d_wave_cycles.p;
    R.heap_dirty = b->d_heap_dirty.p; R.image_words = b->heap_image_words; R.L = b->L;
    R.commit_out = b->d_commit.p; R.dq_count = b->d_dq_count.p; R.n_instances = b->n;
    R.storage_slots = b->lim.storage_slots; R.storage_dirty = b->d_storage_dirty.p;
    R.history = (uint4*)b->d_history.p; R.history16 = (uint32_t)(b->d_history.bytes() / 16);
    HIP_TRY(c, ensure(b->d_reset_params, 1));
    HIP_TRY(c, hipMemcpy(b->d_reset_params.p, &R, sizeof R, hipMemcpyHostToDevice));
    const uint32_t caps[3] = {b->cap_mem, b->cap_log, b->cap_aux};
    const uint32_t per_inst[3] = {b->lim.max_mem_queries, b->lim.max_log_queries, b->lim.max_aux_events};
    const uint32_t max_cap = std::max(caps[0], std::max(caps[1], caps[2]));
    const uint32_t max_per = std::max(per_inst[0], std::max(per_inst[1], per_inst[2]));
    (void)max_cap; (void)max_per;
    // per-instance index lists and counts of every queue side by side: the queues of a step are chained in ONE launch
    // (the lists of a wave share its stream capacity — zkw_commit_params.pooled: an instance that logged more than its nominal
    // share without overflowing the wave's stream is committed in full, as its trace is returned in full)
    const size_t n_padded = (size_t)b->n_waves * b->L;
    HIP_TRY(c, ensure(b->d_idx, n_padded * ((size_t)per_inst[0] + per_inst[1] + per_inst[2])));
    HIP_TRY(c, ensure(b->d_counts, (size_t)b->n * ZKW_QUEUE_COUNT * 2));
    const uint4* streams[3] = {b->d_mem.p, b->d_log.p, b->d_auxs.p};
    zkw_commit_params CP[ZKW_QUEUE_COUNT];
    std::memset(CP, 0, sizeof CP);
    for (uint32_t q = 0; q < ZKW_QUEUE_COUNT; q++) {
      zkw_commit_params& C = CP[q];
      C.n_instances = b->n; C.L = b->L; C.n_waves = b->n_waves; C.max_cycles = b->lim.max_cycles; C.wave_threads = (uint32_t)c->wave_width;
      C.queue = q; C.cap = caps[q]; C.per_instance_cap = per_inst[q]; C.n_blobs = (uint32_t)b->blobs.size();
      C.rc = b->d_rc.p; C.stream = streams[q]; C.cursors = b->d_cursors.p; C.dir = b->d_dir.p; C.scalars = b->d_scalars.p;
      C.blob_digests = b->d_blob_digests.p; C.blob_dir = b->d_blob_dir.p; C.leaves = nullptr;
      C.idx = b->d_idx.p + n_padded * (q == 0 ? 0 : (q == 1 ? per_inst[0] : (size_t)per_inst[0] + per_inst[1])); C.counts = b->d_counts.p + (size_t)b->n * q;
      C.pooled = 1; C.offs = b->d_counts.p + (size_t)b->n * (ZKW_QUEUE_COUNT + q);
      C.out = b->d_commit.p; C.midstates = b->d_midstates.p; C.preimages = b->d_preimages.p; C.n_preimages = (uint32_t)b->preimages.size();
    }
    HIP_TRY(c, ensure(b->d_commit_params, ZKW_QUEUE_COUNT + 1));
    HIP_TRY(c, hipMemcpy(b->d_commit_params.p, CP, sizeof CP, hipMemcpyHostToDevice));
  }
  {  // the host side of the inputs, as the rebuild needs them (BatchInputs)
    auto code = std::make_shared<CodeInputs>();
    code->pages.resize(b->n);
    for (uint32_t i = 0; i < b->n; i++) code->pages[i] = b->staged[i].code_pages;
    code->frames0.resize(b->n);
    for (uint32_t i = 0; i < b->n; i++)
      for (const zkw_callstack_entry& e : b->staged[i].inner) code->frames0[i].emplace_back(e.base_memory_page, e.code_page);
    code->blobs = b->blobs;
    code->preimage_blob.reserve(b->preimages.size());
    for (const auto& pr : b->preimages) code->preimage_blob.push_back(pr.second);
    code->time_delta = c->isa.consts.time_delta_per_cycle;
    auto in = std::make_shared<BatchInputs>();
    in->states.resize(b->n);
    for (uint32_t i = 0; i < b->n; i++) in->states[i] = b->staged[i].state;
    in->code = code;
    {
      auto hv = std::make_shared<std::vector<zkw_u256>>((size_t)b->n * b->heap_image_words);
      for (uint32_t i = 0; i < b->n; i++) {
        const auto& h = b->staged[i].heap;
        if (!h.empty()) std::memcpy(hv->data() + (size_t)i * b->heap_image_words, h.data(), std::min<size_t>(h.size(), b->heap_image_words) * 32);
      }
      in->heap_data = hv->data();
      in->heap_owner = hv;
      in->heap_words = b->heap_image_words;
      in->heaps_known = true;  // (an upload is refused while the staged heaps are older than a restage's)
    }
    b->inputs = in;
  }
  b->uploaded = true;
  b->ran = false;
  return zkw_batch_reset(b, nullptr);
}

// ---- enqueue helpers: every launch covers a list of batches of one context (grid.y / grid.z = batch) ----
static int check_group(zkw_batch* const* bs, uint32_t n) {
  if (!bs || n == 0 || !bs[0]) return ZKW_ERR_INVALID;
  zkw_ctx* c = bs[0]->ctx;
  if (n > ZKW_MAX_FUSED) {
    c->last_error = "more than ZKW_MAX_FUSED batches in one fused step";
    return ZKW_ERR_LIMIT;
  }
  for (uint32_t i = 0; i < n; i++) {
    if (!bs[i] || bs[i]->ctx != c) {
      c->last_error = "fused batches must belong to one context";
      return ZKW_ERR_INVALID;
    }
    if (!bs[i]->uploaded) {
      c->last_error = "zkw_batch_upload first";
      return ZKW_ERR_INVALID;
    }
    for (uint32_t j = 0; j < i; j++)
      if (bs[j] == bs[i]) {
        c->last_error = "a batch appears twice in one fused step";
        return ZKW_ERR_INVALID;
      }
  }
  return ZKW_OK;
}

// host side of a reset (the device side is zkw_reset_kernel)
static void reset_bookkeeping(zkw_batch* const* bs, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    zkw_batch* b = bs[i];
    b->cycles_run = 0;
    b->dq_mode = 0;
    b->ran = false;
    b->synced = false;
    b->ns_done = false;
    b->ns_cached_wave = 0xffffffffu;
    b->wave_cache.clear();
  }
}

static int enqueue_reset(zkw_batch* const* bs, uint32_t n, hipStream_t st) {
  zkw_ctx* c = bs[0]->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  zkw_fused_table T;
  std::memset(&T, 0, sizeof T);
  T.n = n;
  T.wave_threads = (uint32_t)c->wave_width;
  for (uint32_t i = 0; i < n; i++) {
    T.p[i] = bs[i]->d_reset_params.p;
    if (bs[i]->full_reset_pending) T.reserved[1] = 1;  // any freshly uploaded batch in the group: whole heap images for all
    bs[i]->full_reset_pending = false;
  }
  T.reserved[2] = c->opt_reset_skip;  // profiling ablation only
  HIP_TRY(c, zkw_launch_reset_kernel(&T, st));
  reset_bookkeeping(bs, n);
  return ZKW_OK;
}

// Waves per workgroup of one launch of the cycle kernel.  A CU holds 8 waves of it (256 registers: two per SIMD) and
// 160 KB of LDS (16 KB of ISA table per workgroup + 14 KB per wave); the waves of all batches of the launch are numbered
// through, so the choice is free per launch: the fewest rounds of workgroups, then the fewest waves on the busiest CU —
// the kernel runs at the pace of its busiest CU (its waves queue for the CU's memory path).  The driver's 20 batches are
// 1280 waves: as workgroups of 4 they were 320 on 256 CUs (64 CUs with 8 waves, 192 with 4); as 256 workgroups of 5
// every CU has 5.
static uint32_t pick_waves_per_group(const zkw_ctx* c, uint32_t total_waves) {
  if (c->wave_width <= 1) return 1;  // emulation build: one thread is one "wave"
  if (c->opt_waves_per_group) return c->opt_waves_per_group;
  const uint32_t cus = (uint32_t)c->n_cus;
  uint32_t best = ZKW_WAVES_PER_GROUP;
  uint64_t best_key = ~0ull;
  for (uint32_t g = ZKW_WAVES_PER_GROUP; g <= ZKW_MAX_WAVES_PER_GROUP; g++) {
    const uint32_t lds = zkw_cycle_kernel_lds_bytes(ZKW_WAVE, g);
    const uint32_t per_cu = std::max(1u, std::min(ZKW_MAX_WAVES_PER_GROUP / g, (160u * 1024u) / lds));
    const uint32_t n_wg = (total_waves + g - 1) / g;
    const uint32_t rounds = (n_wg + cus * per_cu - 1) / (cus * per_cu);
    const uint32_t wg_last = n_wg - (rounds - 1) * cus * per_cu;
    const uint32_t busiest = ((wg_last + cus - 1) / cus) * g;
    const uint64_t key = ((uint64_t)rounds << 32) | ((uint64_t)busiest << 8) | g;  // ties: the smaller workgroup
    if (key < best_key) {
      best_key = key;
      best = g;
    }
  }
  return best;
}

// `inline_decommit`: the cycle kernel chains the decommit-queue commitment itself (a step that runs and commits in one
// call has nothing to overlap the commitment kernels with); otherwise zkw_batch_commit computes it from the aux stream
static int enqueue_run(zkw_batch* const* bs, uint32_t n, uint32_t max_cycles, hipStream_t st, bool inline_decommit = false, bool whole_step = false) {
  zkw_ctx* c = bs[0]->ctx;
  if (c->opt_no_inline_decommit) inline_decommit = false;  // A/B switch
  for (uint32_t i = 0; i < n; i++) {
    const int want = inline_decommit ? 1 : 2;
    if (bs[i]->dq_mode == 0) bs[i]->dq_mode = want;
    else if (bs[i]->dq_mode != want) inline_decommit = bs[i]->dq_mode == 1;  // a continued run keeps the mode of its first launch
  }
  for (uint32_t i = 0; i < n; i++)
    if (bs[i]->dq_mode != (inline_decommit ? 1 : 2)) {
      c->last_error = "fused run: the batches disagree on how their decommit queue is committed since their last reset";
      return ZKW_ERR_INVALID;
    }
  for (uint32_t i = 0; i < n; i++)
    if (max_cycles == 0 || (uint64_t)bs[i]->cycles_run + max_cycles > bs[i]->lim.max_cycles) {
      c->last_error = "run exceeds limits.max_cycles since the last reset";
      return ZKW_ERR_LIMIT;
    }
  HIP_TRY(c, hipSetDevice(c->device));
  zkw_launch_args A;
  std::memset(&A, 0, sizeof A);
  A.n_batches = n;
  A.run_cycles = max_cycles;
  A.wave_threads = bs[0]->kp.wave_threads;
  bool uniform = true;
  for (uint32_t i = 0; i < n; i++) {
    A.kp[i] = bs[i]->d_kp.p;
    A.wave_base[i + 1] = A.wave_base[i] + bs[i]->n_waves;
    uniform = uniform && bs[i]->n_waves == bs[0]->n_waves;
    A.max_waves = std::max(A.max_waves, bs[i]->n_waves);
    A.max_L = std::max(A.max_L, bs[i]->L);
  }
  A.uniform_waves = uniform ? bs[0]->n_waves : 0;
  A.waves_per_group = pick_waves_per_group(c, A.wave_base[n]);
  // profiling ablations / test hooks only; the launch-internal bits (inline decommit chain, helper waves) are decided
  // below and never taken from the caller: set from outside with no helper wave launched they would make the cycle waves
  // post into LDS address 0 (the ISA table) or wait for a helper that does not exist
  A.debug_flags = c->opt_debug_flags & ~(16u | ZKW_DQ_HELPER | ZKW_KECCAK_HELPER);
  if (inline_decommit) A.debug_flags |= 16u;
  {
    // A helper wave per workgroup takes the decommit chain off the cycle waves (zkw_dq_helper) when the CUs have a wave
    // slot to spare: one more wave per workgroup must not cost a round of workgroups (8 waves of this kernel per CU)
    const uint32_t g = A.waves_per_group, n_wg = (A.wave_base[n] + g - 1) / g;
    const uint32_t lds = zkw_cycle_kernel_lds_bytes(ZKW_WAVE, g) + g * ZKW_DQ_HELPER_BYTES;
    const uint32_t per_cu = g + 1 <= ZKW_MAX_WAVES_PER_GROUP ? std::min(ZKW_MAX_WAVES_PER_GROUP / (g + 1), (160u * 1024u) / lds) : 0u;
    // ... and only inside a whole step (zkw_batches_step: reset, run and commitments of one group on one stream).  A caller
    // that runs and commits separately does so to overlap the commitment kernels of one group with the cycle kernel of
    // another (bench.py's default command): there the "spare" slots are what those kernels run in — with helper waves
    // cfg 4 with all three commitments fell from 1.77 to 1.28 G cycles/s (profiles/r04_ab_settle.txt).
    if (whole_step && inline_decommit && c->wave_width > 1 && !(c->opt_debug_flags & ZKW_NO_DQ_HELPER) && per_cu && n_wg <= (uint32_t)c->n_cus * per_cu) {
      A.helpers = 1;
      A.debug_flags |= ZKW_DQ_HELPER;
    }
  }
  // Batches of thin waves (<= 8 instances per wave: a caller after latency, not throughput): every cycle wave gets helper
  // waves of its own that run its keccak256 calls lane-parallel, one call per helper at a time (zkw_kh_helper) and, inside a
  // whole step, chain its decommits.  Workgroups of g cycle + g * n_sub helper waves: as many helpers per cycle wave as it
  // has lanes (up to 4), the smallest g, as long as the launch stays one round of workgroups.
  if (c->wave_width > 1 && A.max_L <= ZKW_KH_MAX_LANES && !c->opt_waves_per_group && !(c->opt_debug_flags & ZKW_NO_DQ_HELPER)) {
    bool done = false;
    for (uint32_t n_sub = std::min(A.max_L, 4u) >= 4 ? 4 : (A.max_L >= 2 ? 2 : 1); n_sub && !done; n_sub >>= 1) {
      for (uint32_t g = 1; g * (1 + n_sub) <= ZKW_MAX_WAVES_PER_GROUP && !done; g++) {
        const uint32_t lds = zkw_cycle_kernel_lds_bytes(ZKW_WAVE, g) + g * (ZKW_DQ_HELPER_BYTES + ZKW_KH_BYTES);
        const uint32_t per_cu = std::min(ZKW_MAX_WAVES_PER_GROUP / (g * (1 + n_sub)), (160u * 1024u) / lds);
        const uint32_t n_wg = (A.wave_base[n] + g - 1) / g;
        if (!per_cu || n_wg > (uint32_t)c->n_cus * per_cu) continue;
        A.waves_per_group = g;
        A.helpers = g * n_sub;
        A.debug_flags |= ZKW_KECCAK_HELPER;
        if (whole_step && inline_decommit) A.debug_flags |= ZKW_DQ_HELPER; else A.debug_flags &= ~ZKW_DQ_HELPER;
        done = true;
      }
    }
  }
  // HIP events around the launch: on the first batch of the group (its kernel_ms is the launch's duration)
  zkw_batch* lead = bs[0];
  const uint32_t slot = lead->pending_runs % zkw_batch::EV_RING;
  HIP_TRY(c, hipEventRecord(lead->evs[2 * slot], st));
  HIP_TRY(c, zkw_launch_cycle_kernel(&A, st));
  HIP_TRY(c, hipEventRecord(lead->evs[2 * slot + 1], st));
  if (c->opt_debug_sync) HIP_TRY(c, hipStreamSynchronize(st));  // diagnostics only
  lead->pending_runs++;
  for (uint32_t i = 0; i < n; i++) {
    zkw_batch* b = bs[i];
    b->run_stream = st;
    b->cycles_run += max_cycles;
    b->ran = true;
    b->synced = false;
    b->ns_done = false;
    b->ns_cached_wave = 0xffffffffu;
    b->wave_cache.clear();
  }
  return ZKW_OK;
}

static int enqueue_commit(zkw_batch* const* bs, uint32_t n, uint32_t queue_mask, hipStream_t st) {
  zkw_ctx* c = bs[0]->ctx;
  for (uint32_t i = 0; i < n; i++)
    if (!bs[i]->ran) return ZKW_ERR_NOT_RUN;
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t todo = 0;
  for (uint32_t q = 0; q < ZKW_QUEUE_COUNT; q++) {
    if (!((queue_mask >> q) & 1u)) continue;
    if (q == ZKW_QUEUE_DECOMMIT) {  // already chained by the cycle kernel (op_far_call) when the run was part of a fused step
      bool all_inline = true;
      for (uint32_t i = 0; i < n; i++) all_inline = all_inline && bs[i]->dq_mode == 1;
      if (all_inline) continue;
      for (uint32_t i = 0; i < n; i++)
        if (bs[i]->dq_mode == 1) {
          c->last_error = "fused commit: some batches chained their decommit queue in the cycle kernel, others did not";
          return ZKW_ERR_INVALID;
        }
    }
    todo |= 1u << q;
  }
  if (!todo) return ZKW_OK;
  // One bucket launch and one chain launch for all requested queues (grid.z = queue): the chains of a queue are sequential
  // per instance, so the queues side by side double / triple the waves that hide each other's latency.  No leaf pass: the
  // chain kernel hashes the records themselves (memory / log) or uses the cached leaf of the code (decommit).
  zkw_fused_table T;
  std::memset(&T, 0, sizeof T);
  T.n = n;
  T.reserved[0] = 0;
  T.reserved[1] = todo;  // queue mask: T.p[i] is the batch's parameter block of queue 0, the kernels index it by queue
  T.wave_threads = (uint32_t)c->wave_width;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t caps[3] = {bs[i]->cap_mem, bs[i]->cap_log, bs[i]->cap_aux};
    T.p[i] = bs[i]->d_commit_params.p;
    T.max_waves = std::max(T.max_waves, bs[i]->n_waves);
    for (uint32_t q = 0; q < ZKW_QUEUE_COUNT; q++)
      if ((todo >> q) & 1u) T.max_cap = std::max(T.max_cap, caps[q]);
  }
  HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_BUCKET, st));
  HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_CHAIN, st));
  return ZKW_OK;
}

int zkw_batch_reset(zkw_batch* b, void* hip_stream) {
  int rc = check_group(&b, 1);
  return rc != ZKW_OK ? rc : enqueue_reset(&b, 1, (hipStream_t)hip_stream);
}

int zkw_batch_run(zkw_batch* b, uint32_t max_cycles, void* hip_stream) {
  int rc = check_group(&b, 1);
  return rc != ZKW_OK ? rc : enqueue_run(&b, 1, max_cycles, (hipStream_t)hip_stream);
}

int zkw_batches_reset(zkw_batch* const* batches, uint32_t n_batches, void* hip_stream) {
  int rc = check_group(batches, n_batches);
  return rc != ZKW_OK ? rc : enqueue_reset(batches, n_batches, (hipStream_t)hip_stream);
}

int zkw_batches_run(zkw_batch* const* batches, uint32_t n_batches, uint32_t max_cycles, void* hip_stream) {
  int rc = check_group(batches, n_batches);
  return rc != ZKW_OK ? rc : enqueue_run(batches, n_batches, max_cycles, (hipStream_t)hip_stream);
}

int zkw_batches_run_committing(zkw_batch* const* batches, uint32_t n_batches, uint32_t max_cycles, uint32_t queue_mask, void* hip_stream) {
  int rc = check_group(batches, n_batches);
  return rc != ZKW_OK ? rc : enqueue_run(batches, n_batches, max_cycles, (hipStream_t)hip_stream, (queue_mask >> ZKW_QUEUE_DECOMMIT) & 1u);
}

int zkw_batches_commit(zkw_batch* const* batches, uint32_t n_batches, uint32_t queue_mask, void* hip_stream) {
  int rc = check_group(batches, n_batches);
  if (rc != ZKW_OK) return rc;
  return queue_mask ? enqueue_commit(batches, n_batches, queue_mask, (hipStream_t)hip_stream) : ZKW_OK;
}

int zkw_batches_step(zkw_batch* const* batches, uint32_t n_batches, uint32_t max_cycles, uint32_t queue_mask, void* hip_stream) {
  int rc = check_group(batches, n_batches);
  if (rc != ZKW_OK) return rc;
  hipStream_t st = (hipStream_t)hip_stream;
  rc = enqueue_reset(batches, n_batches, st);
  if (rc == ZKW_OK) rc = enqueue_run(batches, n_batches, max_cycles, st, (queue_mask >> ZKW_QUEUE_DECOMMIT) & 1u, true);
  if (rc == ZKW_OK && queue_mask) rc = enqueue_commit(batches, n_batches, queue_mask, st);
  return rc;
}

int zkw_batches_step_prepared(zkw_batch* const* batches, uint32_t n_batches, uint32_t max_cycles, uint32_t queue_mask, void* hip_stream) {
  int rc = check_group(batches, n_batches);
  if (rc != ZKW_OK) return rc;
  hipStream_t st = (hipStream_t)hip_stream;
  rc = enqueue_run(batches, n_batches, max_cycles, st, (queue_mask >> ZKW_QUEUE_DECOMMIT) & 1u, true);
  if (rc == ZKW_OK && queue_mask) rc = enqueue_commit(batches, n_batches, queue_mask, st);
  return rc;
}

// mean device time of the cycle-kernel launches recorded on this batch since the last drain (HIP event pairs)
static int drain_timing(zkw_batch* b) {
  zkw_ctx* c = b->ctx;
  if (!b->pending_runs) return ZKW_OK;
  const uint32_t last = (b->pending_runs - 1) % zkw_batch::EV_RING;
  HIP_TRY(c, hipEventSynchronize(b->evs[2 * last + 1]));
  const uint32_t cnt = std::min<uint32_t>(b->pending_runs, zkw_batch::EV_RING);
  float total = 0;
  for (uint32_t i = 0; i < cnt; i++) {
    float ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms, b->evs[2 * i], b->evs[2 * i + 1]));
    total += ms;
  }
  b->kernel_ms = total / cnt;
  b->timed_runs = cnt;
  b->pending_runs = 0;
  return ZKW_OK;
}

int zkw_batch_kernel_time(zkw_batch* b, double* mean_ms, uint32_t* n_launches) {
  if (!b || !mean_ms) return ZKW_ERR_INVALID;
  if (!b->uploaded) return ZKW_ERR_INVALID;
  HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
  int rc = drain_timing(b);
  if (rc != ZKW_OK) return rc;
  *mean_ms = b->kernel_ms;
  if (n_launches) *n_launches = b->timed_runs;
  return ZKW_OK;
}

int zkw_batch_sync(zkw_batch* b) {
  if (!b) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->ran) {
    c->last_error = "nothing was run";
    return ZKW_ERR_NOT_RUN;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  {
    int trc = drain_timing(b);
    if (trc != ZKW_OK) return trc;
  }
  HIP_TRY(c, hipStreamSynchronize(b->run_stream));
  b->h_scalars.resize(b->n);
  HIP_TRY(c, hipMemcpy(b->h_scalars.data(), b->d_scalars.p, b->d_scalars.bytes(), hipMemcpyDeviceToHost));
  b->h_cursors.resize((size_t)b->n_waves * 4);
  HIP_TRY(c, hipMemcpy(b->h_cursors.data(), b->d_cursors.p, b->d_cursors.bytes(), hipMemcpyDeviceToHost));
  b->synced = true;
  return ZKW_OK;
}

// Bulk download of everything a run produced (tails, register deltas, query streams, directory) into pinned host
// staging buffers: what a consumer on the host side of PCIe would have to pull.  Only the used extents travel.
int zkw_batch_download_all(zkw_batch* b, uint64_t* n_bytes, double* ms) {
  if (!b || !n_bytes) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->synced) {
    int rc = zkw_batch_sync(b);
    if (rc != ZKW_OK) return rc;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t W = b->n_waves, L = b->L, MC = b->lim.max_cycles;
  uint32_t max_cyc = 0;
  for (uint32_t i = 0; i < b->n; i++) max_cyc = std::max(max_cyc, b->h_scalars[i].n_cycles);
  struct Piece { const void* src; size_t pitch, bytes; };
  std::vector<Piece> pieces;
  size_t total = 0;
  for (uint32_t w = 0; w < W; w++) {
    const uint32_t* cur = &b->h_cursors[(size_t)w * 4];
    pieces.push_back({b->d_tails.p + (size_t)w * MC * L, 0, (size_t)max_cyc * L * 16});
    pieces.push_back({b->d_deltas.p + (size_t)w * b->cap_delta * 2, 0, (size_t)std::min(cur[3], b->cap_delta) * 16});                   // low plane
    pieces.push_back({b->d_deltas.p + (size_t)w * b->cap_delta * 2 + b->cap_delta, 0, (size_t)std::min(cur[3], b->cap_delta) * 16});  // high plane
    for (int pl = 0; pl < 3; pl++)  // three planes
      pieces.push_back({b->d_mem.p + ((size_t)w * 3 + pl) * b->cap_mem, 0, (size_t)std::min(cur[0], b->cap_mem) * 16});
    pieces.push_back({b->d_log.p + (size_t)w * b->cap_log * 8, 0, (size_t)std::min(cur[1], b->cap_log) * 128});
    pieces.push_back({b->d_auxs.p + (size_t)w * b->cap_aux * 16, 0, (size_t)std::min(cur[2], b->cap_aux) * 256});
    pieces.push_back({b->d_dir.p + (size_t)w * (MC + 1) * 4, 0, (size_t)(max_cyc + 1) * 16});
  }
  for (const Piece& p : pieces) total += p.bytes;
  void* host = nullptr;
  HIP_TRY(c, hipHostMalloc(&host, std::max<size_t>(total, 16), hipHostMallocDefault));
  hipEvent_t e0, e1;
  HIP_TRY(c, hipEventCreate(&e0));
  HIP_TRY(c, hipEventCreate(&e1));
  hipStream_t st = b->run_stream;
  HIP_TRY(c, hipEventRecord(e0, st));
  size_t off = 0;
  for (const Piece& p : pieces) {
    if (p.bytes) HIP_TRY(c, hipMemcpyAsync((char*)host + off, p.src, p.bytes, hipMemcpyDeviceToHost, st));
    off += p.bytes;
  }
  HIP_TRY(c, hipEventRecord(e1, st));
  HIP_TRY(c, hipEventSynchronize(e1));
  float t = 0;
  HIP_TRY(c, hipEventElapsedTime(&t, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipHostFree(host);
  *n_bytes = total;
  if (ms) *ms = t;
  return ZKW_OK;
}


int zkw_batch_get_stats(zkw_batch* b, zkw_run_stats* out) {
  if (!b || !out) return ZKW_ERR_INVALID;
  if (!b->synced) {
    int rc = zkw_batch_sync(b);
    if (rc != ZKW_OK) return rc;
  }
  std::memset(out, 0, sizeof *out);
  for (uint32_t i = 0; i < b->n; i++) {
    const zkw_dev_scalars& s = b->h_scalars[i];
    out->cycles += s.n_cycles;
    if (s.status == ZKW_STATUS_ENDED) out->instances_ended++;
    if (s.status >= ZKW_STATUS_UNKNOWN_CODE_HASH) out->instances_failed++;
  }
  for (uint32_t w = 0; w < b->n_waves; w++) {  // stream totals (include the discarded records of failed cycles)
    out->mem_queries += std::min(b->h_cursors[(size_t)w * 4 + 0], b->cap_mem);
    out->log_queries += std::min(b->h_cursors[(size_t)w * 4 + 1], b->cap_log);
    out->aux_events += std::min(b->h_cursors[(size_t)w * 4 + 2], b->cap_aux);
    out->reg_deltas += std::min(b->h_cursors[(size_t)w * 4 + 3], b->cap_delta);
  }
  out->kernel_ms = b->kernel_ms;
  return ZKW_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Host rebuild of a wave's witness from the link format (zkw_pack.h): ONE walk over the packed streams of a wave, cycle by
// cycle, that keeps the current 512-byte snapshot of every lane (initial register file + the deltas applied so far), gives
// every lane its queries of the cycle in emission order, and hands both to a sink.  Two sinks: the materialising one behind
// zkw_batch_get_instance_trace / zkw_delivery_get_instance_trace (per-instance arrays, as the C ABI returns them) and the
// streaming one behind zkw_delivery_replay (a callback per instance and cycle with a pointer to the live snapshot — what
// start_new_execution_cycle(&local_state) / end_execution_cycle(&local_state) are in the reference, witness_trace/mod.rs:11-20:
// a reference to the VM's state, not a copy per cycle).  Read-only on the batch: any number of threads may walk different
// waves at once.
// ---------------------------------------------------------------------------------------------------------------------
struct WaveView {  // host pointers to the packed data of one wave
  uint32_t flags = 0;  // ZKW_PACK_* of the block
  uint32_t L = 0, max_cyc = 0, n_delta = 0, n_mem = 0, n_val = 0, n_log = 0, n_aux = 0, aux_units = 0, n_page = 0;
  const uint32_t* dir = nullptr;
  const uint4 *tails = nullptr, *dlo = nullptr, *dhi = nullptr;
  const uint32_t *tx = nullptr, *ty = nullptr, *tz = nullptr;  // slim tails (ZKW_PACK_SLIM_TAILS): planes x | y | z ...
  const uint8_t* tb = nullptr;                                 // ... and the high byte of the delta mask
  const uint32_t *tw = nullptr, *lx = nullptr, *ly = nullptr, *lz = nullptr;  // tails as differences (ZKW_PACK_DELTA_TAILS): zkw_pack_tail_word + the three lists
  uint32_t n_tx = 0, n_ty = 0, n_tz = 0;
  // the 16-byte tail of lane-cycle i as the cycle kernel stored it, but for the event counts (not on the link when slim: tail_counts)
  uint4 tail(size_t i) const { return tails ? tails[i] : make_uint4(tx[i], ty[i], tz[i], (uint32_t)tb[i] << 24); }
  uint32_t delta_mask(size_t i) const {
    if (tw) return tw[i] & 0xffffu;
    const uint4 t0 = tail(i);
    return (t0.x >> 24) | ((t0.w >> 24) << 8);
  }
  // sparse deltas (ZKW_PACK_SPARSE_DELTAS, zkw_pack_delta_units): the bit planes, bytes 0..7 of every delta, bytes 8..15 / 16..31 of those that have any
  const uint64_t *dbits = nullptr, *d0 = nullptr, *d1 = nullptr;
  const uint4* d2 = nullptr;
  uint32_t n_d1 = 0, n_d2 = 0;
  bool delta_has1(uint32_t i) const { return (dbits[8u * (i >> 8) + (i & 3u)] >> ((i & 255u) >> 2)) & 1u; }
  bool delta_has2(uint32_t i) const { return (dbits[8u * (i >> 8) + 4u + (i & 3u)] >> ((i & 255u) >> 2)) & 1u; }
  const uint32_t *m_page = nullptr, *m_index = nullptr, *m_misc = nullptr;
  const uint4 *v_lo = nullptr, *v_hi = nullptr, *log = nullptr, *aux = nullptr;
};

static bool wave_view(const uint4* block, uint64_t block_units, const zkw_pack_wave& e, uint32_t L, uint32_t flags, WaveView& v) {
  if (e.off == 0 || (uint64_t)e.off + e.units > block_units) return false;  // (not packed, or an extent that is not inside the block)
  v.flags = flags;
  v.L = L; v.max_cyc = e.max_cyc; v.n_delta = e.n_delta; v.n_mem = e.n_mem; v.n_val = e.n_val; v.n_log = e.n_log; v.n_aux = e.n_aux; v.aux_units = e.aux_units;
  v.n_page = e.n_page;
  zkw_pack_counts C;
  C.max_cyc = e.max_cyc; C.L = L; C.n_delta = e.n_delta; C.n_mem = e.n_mem; C.n_page = e.n_page; C.n_val = e.n_val; C.n_log = e.n_log; C.aux_units = e.aux_units;
  C.n_d1 = e.n_d1; C.n_d2 = e.n_d2; C.n_tx = e.n_tx; C.n_ty = e.n_ty; C.n_tz = e.n_tz;
  v.n_d1 = e.n_d1; v.n_d2 = e.n_d2; v.n_tx = e.n_tx; v.n_ty = e.n_ty; v.n_tz = e.n_tz;
  if (zkw_pack_wave_units(&C, flags) != e.units) return false;  // (the entry does not describe its own extent: nothing of it can be trusted)
  const uint4* d = block + e.off;
  v.dir = (const uint32_t*)d; d += e.max_cyc + 1;
  {
    const uint64_t n_t = (uint64_t)e.max_cyc * L;
    if (flags & ZKW_PACK_DELTA_TAILS) {
      const uint64_t t4 = zkw_ceil4_64(n_t);
      v.tails = nullptr;
      v.tw = (const uint32_t*)d;
      v.lx = (const uint32_t*)(d + t4); v.ly = (const uint32_t*)(d + t4 + zkw_ceil4_64(e.n_tx)); v.lz = (const uint32_t*)(d + t4 + zkw_ceil4_64(e.n_tx) + zkw_ceil4_64(e.n_ty));
    } else if (flags & ZKW_PACK_SLIM_TAILS) {
      const uint64_t t4 = (n_t + 3) >> 2;
      v.tails = nullptr;
      v.tx = (const uint32_t*)d; v.ty = (const uint32_t*)(d + t4); v.tz = (const uint32_t*)(d + 2 * t4); v.tb = (const uint8_t*)(d + 3 * t4);
    } else {
      v.tails = d;
    }
    d += zkw_pack_tail_units(n_t, &C, flags);
  }
  if (flags & ZKW_PACK_SPARSE_DELTAS) {
    const uint64_t nblk = ((uint64_t)e.n_delta + 255) >> 8;
    v.dlo = v.dhi = nullptr;
    v.dbits = (const uint64_t*)d;
    v.d0 = (const uint64_t*)(d + 4 * nblk);
    v.d1 = (const uint64_t*)(d + 4 * nblk + zkw_ceil2_64(e.n_delta));
    v.d2 = d + 4 * nblk + zkw_ceil2_64(e.n_delta) + zkw_ceil2_64(e.n_d1);
    d += zkw_pack_delta_units(&C, flags);
  } else {
    v.dlo = d; v.dhi = d + e.n_delta; d += 2 * (size_t)e.n_delta;
  }
  const uint32_t q4 = zkw_ceil4(e.n_mem);
  v.m_page = (const uint32_t*)d; d += zkw_ceil4(e.n_page);  // the page list: the queries that carry their page, in stream order
  v.m_index = (const uint32_t*)d; v.m_misc = (const uint32_t*)(d + q4); d += 2 * (size_t)q4;
  v.v_lo = d; v.v_hi = d + e.n_val; d += 2 * (size_t)e.n_val;
  v.log = d; d += (size_t)e.n_log * 8;
  v.aux = d;
  return true;
}

// what a sink sees of one instance in one cycle; the arrays live until the next call
struct CycleView {
  uint32_t lane, cycle;
  const zkw_cycle_record* record;
  const zkw_mem_query* mem; uint32_t n_mem;
  const zkw_log_query* log; uint32_t n_log;
  const zkw_aux_event* aux; uint32_t n_aux;
};

// The containers of walk_wave: a replay thread walks thousands of waves, and what a wave needs — the lanes' live records, their
// shadow pages, the per-cycle buckets — keeps its capacity from one wave to the next instead of going through malloc / free per
// wave and lane (an eighth of the walk's time, profiles/r10_host_replay.txt).
struct WalkShadowPage {
  uint32_t page = 0;
  const zkw_u256* image = nullptr;  // the staged heap image of this page (copied into `w` when the page is first touched)
  uint32_t image_words = 0;
  std::vector<zkw_u256> w;          // dense; words at and beyond w.size() are zero
};
struct WalkShadow {  // the pages one lane has touched
  std::vector<WalkShadowPage> pages;  // (the first `n` are in use: the others keep their buffers for the next wave)
  uint32_t n = 0, last = 0;
  void reset() { n = 0; last = 0; }
  WalkShadowPage* add(uint32_t page) {
    if (n == pages.size()) pages.emplace_back();
    WalkShadowPage* p = &pages[n];
    last = n++;
    p->page = page; p->image = nullptr; p->image_words = 0;
    p->w.clear();
    return p;
  }
  WalkShadowPage* find(uint32_t page) {
    if (last < n && pages[last].page == page) return &pages[last];
    for (uint32_t i = 0; i < n; i++)
      if (pages[i].page == page) { last = i; return &pages[i]; }
    return nullptr;
  }
  static void materialise(WalkShadowPage* p) {
    if (p->image) {
      p->w.assign(p->image, p->image + p->image_words);
      p->image = nullptr;
    }
  }
  void read(uint32_t page, uint32_t index, zkw_u256* out) {
    WalkShadowPage* p = find(page);
    if (p) {
      materialise(p);
      if (index < p->w.size()) { *out = p->w[index]; return; }
    }
    std::memset(out, 0, sizeof *out);
  }
  void write(uint32_t page, uint32_t index, const zkw_u256& val) {
    WalkShadowPage* p = find(page);
    if (!p) p = add(page);
    materialise(p);
    if (index >= p->w.size()) {
      zkw_u256 zero;
      std::memset(&zero, 0, sizeof zero);
      p->w.resize(std::max<size_t>((size_t)index + 1, p->w.size() * 2), zero);
    }
    p->w[index] = val;
  }
};
struct WalkSlow { uint32_t heap_bound, aux_bound, depth, timestamp, pc; };
struct WalkScratch {
  std::vector<zkw_aux_event> aux, ca;
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> pages, frames;
  std::vector<zkw_cycle_record> cur;
  std::vector<WalkSlow> slow;
  std::vector<WalkShadow> shadow;
  std::vector<uint32_t> mask, cnt_m, cnt_l, cnt_a, fill;
  std::vector<zkw_mem_query> cm;
  std::vector<zkw_log_query> cl;
};

template <class Sink>
static void walk_wave(WalkScratch& S, const BatchInputs& in, uint32_t w, const WaveView& v, const uint32_t* ncyc, Sink& sink) {
  const CodeInputs& code = *in.code;
  const uint32_t n_inst = (uint32_t)in.states.size();
  const uint32_t L = v.L;
  uint32_t max_cycles_lane = 0;
  for (uint32_t l = 0; l < L; l++) max_cycles_lane = std::max(max_cycles_lane, ncyc[l]);
  max_cycles_lane = std::min(max_cycles_lane, v.max_cyc);
  // aux records sit back to back, each as long as its type uses: expand them once (zeros behind the used part, as the ABI has it)
  std::vector<zkw_aux_event>& aux = S.aux;
  aux.resize(v.n_aux);
  {
    const uint4* a = v.aux;
    for (uint32_t i = 0; i < v.n_aux; i++) {
      const uint32_t used = zkw_aux_used_units(a->x & 0xffu);
      std::memset(&aux[i], 0, sizeof(zkw_aux_event));
      std::memcpy(&aux[i], a, (size_t)used * 16);
      a += used;
    }
  }
  // code pages of every lane: what the host staged + what the run decommitted (page -> blob; a page number is never reused).
  // A Code query's value does not travel (zkw_pack.h): it is word `index` of the page's blob, zero beyond its length
  // (memory.rs:556-569 read_code_query on a page populated by populate_code / the decommitter, decommitter.rs:81-96).
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>>& pages = S.pages;
  if (pages.size() < L) pages.resize(L);
  for (uint32_t l = 0; l < L; l++) {
    pages[l].clear();
    const uint32_t inst = w * L + l;
    if (inst >= n_inst) continue;
    for (const auto& pg : code.pages[inst]) pages[l].push_back(pg);  // (later registrations win: searched from the back)
  }
  // (the blob of a decommit through its preimage index: the blob id in `c` is the 16-bit field of the reference's decommit query)
  for (uint32_t i = 0; i < v.n_aux; i++)
    if (aux[i].type == ZKW_AUX_DECOMMIT && aux[i].lane < L) {
      const uint32_t pre = aux[i].u.decommit.preimage_index;
      pages[aux[i].lane].emplace_back(aux[i].b, pre < code.preimage_blob.size() ? code.preimage_blob[pre] : 0u);
    }
  auto code_word = [&](uint32_t l, uint32_t page, uint32_t index, zkw_u256* out) {
    std::memset(out, 0, sizeof *out);
    const auto& pv = pages[l];
    for (size_t k = pv.size(); k-- > 0;)
      if (pv[k].first == page) {
        const auto& blob = code.blobs[pv[k].second < code.blobs.size() ? pv[k].second : 0];
        if (index < blob.size()) *out = blob[index];
        return;
      }
  };
  // the live snapshots: registers from the staged initial state, the slow tail fields beside them
  typedef WalkSlow Slow;
  std::vector<zkw_cycle_record>& cur = S.cur;
  std::vector<Slow>& slow = S.slow;
  cur.resize(L);
  slow.resize(L);
  const uint32_t time_delta = code.time_delta;
  for (uint32_t l = 0; l < L; l++) {
    std::memset(&cur[l], 0, sizeof(zkw_cycle_record));
    const uint32_t inst = w * L + l;
    if (inst < n_inst) {
      const zkw_vm_local_state& st0 = in.states[inst];
      std::memcpy(cur[l].registers, st0.registers, sizeof st0.registers);
      slow[l] = Slow{st0.current.heap_bound, st0.current.aux_heap_bound, st0.callstack_depth, st0.timestamp, st0.current.pc};
    } else {
      slow[l] = Slow{0, 0, 0, 0, 0};
    }
  }
  // Shadow memory (blocks packed with ZKW_PACK_NO_READ_VALUES): what a read returns is what was written there before — the
  // heap image the instance was staged with (page base + 2 of the frame it started in: zkw_batch_set_heap), zero on every other
  // page (SimpleMemory hands out zero-filled pages, memory.rs:15-148; `.get(index).unwrap_or(zero)` :490-495), then every write
  // query of the lane in stream order.  A page is a dense array grown on demand (zero-filled); the page that holds the staged heap
  // image starts as a copy of it, made when the lane first touches the page.
  const bool shadowed = (v.flags & ZKW_PACK_NO_READ_VALUES) != 0;
  std::vector<WalkShadow>& shadow = S.shadow;
  if (shadowed) {
    if (shadow.size() < L) shadow.resize(L);
    for (uint32_t l = 0; l < L; l++) {
      shadow[l].reset();
      const uint32_t inst = w * L + l;
      if (inst >= n_inst || !in.heap_data || !in.heap_words) continue;
      WalkShadowPage* pg = shadow[l].add(in.states[inst].current.base_memory_page + 2u);  // heap_page_from_base of the frame the image was staged for
      pg->image = in.heap_data + (size_t)inst * in.heap_words;
      pg->image_words = in.heap_words;
    }
  }
  // Implied pages (ZKW_PACK_IMPLIED_PAGES): the callstack of every lane as (base page, code page) — the inner entries it was staged
  // with, its current entry, then the FRAME_START / FRAME_FINISH events of the aux stream applied at the END of their cycle (a frame
  // changes behind the cycle's last memory query: far_call.rs:562, ret.rs:196-243, near_call.rs:60-67 come after the operand reads).
  const bool implied = (v.flags & ZKW_PACK_IMPLIED_PAGES) != 0;
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>>& frames = S.frames;
  if (implied && frames.size() < L) frames.resize(L);
  if (implied)
    for (uint32_t l = 0; l < L; l++) {
      frames[l].clear();
      const uint32_t inst = w * L + l;
      if (inst >= n_inst) continue;
      if (inst < code.frames0.size()) frames[l].assign(code.frames0[inst].begin(), code.frames0[inst].end());
      frames[l].emplace_back(in.states[inst].current.base_memory_page, in.states[inst].current.code_page);
    }
  const bool sparse = (v.flags & ZKW_PACK_SPARSE_DELTAS) != 0;
  uint32_t dseen = 0, p1 = 0, p2 = 0;  // sparse deltas: deltas passed, entries of the two sparse planes passed
  const bool dtails = v.tw != nullptr;
  uint32_t qx = 0, qy = 0, qz = 0;     // tails as differences: entries of the three lists passed
  uint32_t ppos = 0;  // next entry of the page list
  std::vector<uint32_t>&mask = S.mask, &cnt_m = S.cnt_m, &cnt_l = S.cnt_l, &cnt_a = S.cnt_a, &fill = S.fill;
  mask.assign(L, 0u); cnt_m.assign(L + 1, 0u); cnt_l.assign(L + 1, 0u); cnt_a.assign(L + 1, 0u); fill.assign(L, 0u);
  std::vector<zkw_mem_query>& cm = S.cm;
  std::vector<zkw_log_query>& cl = S.cl;
  std::vector<zkw_aux_event>& ca = S.ca;
  uint32_t vpos = 0;  // next entry of the value planes (the queries that are no Code reads, in stream order)
  uint32_t pm = 0;    // next memory query of the stream
  for (uint32_t k = 0; k < max_cycles_lane; k++) {
    const uint32_t* d0 = v.dir + (size_t)k * 4;
    const uint32_t* d1 = v.dir + (size_t)(k + 1) * 4;
    // ---- the cycle's register deltas: by mask bit (ascending), lanes in lane order within a bit (zkw_cycle_kernel) ----
    uint32_t any = 0;
    for (uint32_t l = 0; l < L; l++) {
      mask[l] = 0;
      if (k >= ncyc[l]) continue;
      mask[l] = v.delta_mask((size_t)k * L + l);
      any |= mask[l];
    }
    uint32_t pos = d0[3];
    if (sparse)  // (the planes of bytes 8..15 / 16..31 are in stream order: skip what no live lane owns)
      for (const uint32_t upto = std::min(pos, v.n_delta); dseen < upto; dseen++) { p1 += v.delta_has1(dseen) ? 1u : 0u; p2 += v.delta_has2(dseen) ? 1u : 0u; }
    for (uint32_t r = 0; r < ZKW_REGISTERS_COUNT + 1 && (any >> r); r++) {
      if (!((any >> r) & 1u)) continue;
      for (uint32_t l = 0; l < L; l++) {
        if (!((mask[l] >> r) & 1u)) continue;
        if (pos < v.n_delta) {
          uint64_t sv[4];  // (r == 15: the slow fields — heap bound | aux-heap bound | depth — travel like a register)
          uint64_t* dst = r < ZKW_REGISTERS_COUNT ? cur[l].registers[r].l : sv;
          if (sparse) {
            for (; dseen < pos; dseen++) { p1 += v.delta_has1(dseen) ? 1u : 0u; p2 += v.delta_has2(dseen) ? 1u : 0u; }
            dst[0] = v.d0[pos];
            if (v.delta_has1(pos)) { dst[1] = p1 < v.n_d1 ? v.d1[p1] : 0; p1++; } else dst[1] = 0;
            if (v.delta_has2(pos)) { if (p2 < v.n_d2) std::memcpy(&dst[2], &v.d2[p2], 16); else dst[2] = dst[3] = 0; p2++; } else dst[2] = dst[3] = 0;
            dseen = pos + 1;
          } else {
            std::memcpy(&dst[0], &v.dlo[pos], 16);
            std::memcpy(&dst[2], &v.dhi[pos], 16);
          }
          if (r >= ZKW_REGISTERS_COUNT) { slow[l].heap_bound = (uint32_t)sv[0]; slow[l].aux_bound = (uint32_t)(sv[0] >> 32); slow[l].depth = (uint32_t)sv[1]; }
        }
        pos++;
      }
    }
    // ---- the cycle's queries, bucketed by lane (stream order preserves each lane's order): count, place ----
    const uint32_t m0 = std::min(d0[0], v.n_mem), m1 = std::min(d1[0], v.n_mem);
    const uint32_t l0 = std::min(d0[1], v.n_log), l1 = std::min(d1[1], v.n_log);
    const uint32_t a0 = std::min(d0[2], v.n_aux), a1 = std::min(d1[2], v.n_aux);
    std::fill(cnt_m.begin(), cnt_m.end(), 0u); std::fill(cnt_l.begin(), cnt_l.end(), 0u); std::fill(cnt_a.begin(), cnt_a.end(), 0u);
    auto live = [&](uint32_t l) { return l < L && k < ncyc[l]; };
    for (uint32_t p = m0; p < m1; p++) { const uint32_t l = v.m_misc[p] & 0xffu; if (live(l)) cnt_m[l + 1]++; }
    for (uint32_t p = l0; p < l1; p++) { const uint32_t l = ((const zkw_log_query*)v.log)[p].lane; if (live(l)) cnt_l[l + 1]++; }
    for (uint32_t p = a0; p < a1; p++) { const uint32_t l = aux[p].lane; if (live(l)) cnt_a[l + 1]++; }
    for (uint32_t l = 0; l < L; l++) { cnt_m[l + 1] += cnt_m[l]; cnt_l[l + 1] += cnt_l[l]; cnt_a[l + 1] += cnt_a[l]; }
    cm.resize(cnt_m[L]); cl.resize(cnt_l[L]); ca.resize(cnt_a[L]);
    // (memory queries of earlier stream positions that belong to no cycle range cannot exist: the ranges tile the stream)
    for (; pm < m0; pm++) {
      if (zkw_pack_has_value(v.m_misc[pm], v.flags)) vpos++;
      if (zkw_pack_has_page(v.m_misc[pm], v.flags)) ppos++;
    }
    std::fill(fill.begin(), fill.end(), 0u);
    for (uint32_t p = m0; p < m1; p++, pm++) {
      const uint32_t misc = v.m_misc[p], l = misc & 0xffu, meta = (misc >> 16) & 0xffu;
      const bool is_code = (meta & ZKW_MQ_TYPE_MASK) == ZKW_MEM_CODE;
      const bool has_value = zkw_pack_has_value(misc, v.flags), has_page = zkw_pack_has_page(misc, v.flags);
      const uint32_t vi = has_value ? vpos++ : 0u;
      const uint32_t pi = has_page ? ppos++ : 0u;
      if (!live(l)) continue;
      zkw_mem_query& q = cm[cnt_m[l] + fill[l]++];
      const uint32_t cycle_ts = slow[l].timestamp;  // the lane's timestamp at the start of this cycle
      q.timestamp = cycle_ts + (((misc >> 24) - cycle_ts) & 0xffu);
      if (has_page) {
        q.page = pi < v.n_page ? v.m_page[pi] : 0u;
      } else {  // a query of the VM itself on a page of its current frame (execution_stack.rs:67-81)
        const uint32_t type = meta & ZKW_MQ_TYPE_MASK;
        const std::pair<uint32_t, uint32_t> top = frames[l].empty() ? std::pair<uint32_t, uint32_t>(0u, 0u) : frames[l].back();
        q.page = type == ZKW_MEM_CODE ? top.second : top.first + (type == ZKW_MEM_STACK ? 1u : type == ZKW_MEM_HEAP ? 2u : 3u);
      }
      q.index = v.m_index[p];
      q.lane = 0; q.seq = (uint8_t)(misc >> 8); q.meta = (uint8_t)meta; q.reserved0 = 0;
      if (is_code) code_word(l, q.page, q.index, &q.value);
      else if (!has_value) shadow[l].read(q.page, q.index, &q.value);  // a read under ZKW_PACK_NO_READ_VALUES
      else if (vi < v.n_val) { std::memcpy(&q.value.l[0], &v.v_lo[vi], 16); std::memcpy(&q.value.l[2], &v.v_hi[vi], 16); }
      else std::memset(&q.value, 0, sizeof q.value);
      if (shadowed && (meta & ZKW_MQ_RW)) shadow[l].write(q.page, q.index, q.value);
    }
    std::fill(fill.begin(), fill.end(), 0u);
    for (uint32_t p = l0; p < l1; p++) {
      const zkw_log_query& src = ((const zkw_log_query*)v.log)[p];
      const uint32_t l = src.lane;
      if (!live(l)) continue;
      zkw_log_query& q = cl[cnt_l[l] + fill[l]++];
      q = src;
      q.lane = 0;
    }
    std::fill(fill.begin(), fill.end(), 0u);
    for (uint32_t p = a0; p < a1; p++) {
      const uint32_t l = aux[p].lane;
      if (!live(l)) continue;
      zkw_aux_event& q = ca[cnt_a[l] + fill[l]++];
      q = aux[p];
      q.lane = 0;
      if (implied) {  // (applied here, behind the cycle's memory queries: they were resolved above)
        if (q.type == ZKW_AUX_FRAME_START) frames[l].emplace_back(q.u.frame.next.base_memory_page, q.u.frame.next.code_page);
        else if (q.type == ZKW_AUX_FRAME_FINISH && !frames[l].empty()) frames[l].pop_back();
      }
    }
    // ---- the tails, then the sink ----
    for (uint32_t l = 0; l < L; l++) {
      uint4 t0;
      if (dtails) {  // (every lane-cycle of the wave has its word: the lists are passed in lane-cycle order, whoever lives)
        const uint32_t tw = v.tw[(size_t)k * L + l];
        const uint32_t ex = zkw_tw_has_x(tw) ? qx++ : 0xffffffffu, ey = zkw_tw_has_y(tw) ? qy++ : 0xffffffffu, ez = zkw_tw_has_z(tw) ? qz++ : 0xffffffffu;
        if (k >= ncyc[l]) continue;
        const uint32_t* pt = (const uint32_t*)&cur[l].tail;  // the lane's previous tail (x | y | z as below)
        t0.x = ex != 0xffffffffu ? (ex < v.n_tx ? v.lx[ex] : 0u) : ((pt[0] & 0xffffu) | (((tw >> 24) & 0xfu) << 16));
        t0.y = ey != 0xffffffffu ? (ey < v.n_ty ? v.ly[ey] : 0u) : (((pt[1] + 1u) & 0xffffu) | (pt[1] & 0xffff0000u));
        t0.z = ez != 0xffffffffu ? (ez < v.n_tz ? v.lz[ez] : 0u) : pt[2] - ((tw >> 16) & 0xffu);
        t0.w = 0;
      } else {
        if (k >= ncyc[l]) continue;
        t0 = v.tail((size_t)k * L + l);
      }
      if (!v.tails) {  // the event counts of the cycle: the numbers of its queries in the three streams, saturating bytes (Lane::counts)
        const uint32_t nm = cnt_m[l + 1] - cnt_m[l], nl = cnt_l[l + 1] - cnt_l[l], na = cnt_a[l + 1] - cnt_a[l];
        t0.w = (t0.w & 0xff000000u) | std::min(nm, 255u) | (std::min(nl, 255u) << 8) | (std::min(na, 255u) << 16);
      }
      const uint32_t super_pc = (slow[l].pc & 0xffffu) >> 2;  // of the pc this cycle started from
      slow[l].timestamp += time_delta;
      slow[l].pc = t0.y & 0xffffu;
      // (the delta mask is device bookkeeping: reserved byte and top byte of the counts are zero in the ABI)
      uint4* tl = (uint4*)&cur[l].tail;
      tl[0] = make_uint4(t0.x & 0x00ffffffu, t0.y, t0.z, slow[l].timestamp);
      tl[1] = make_uint4(slow[l].heap_bound, slow[l].aux_bound, (slow[l].depth & 0xffffu) | (super_pc << 16), t0.w & 0x00ffffffu);
      CycleView cv;
      cv.lane = l; cv.cycle = k; cv.record = &cur[l];
      cv.mem = cm.data() + cnt_m[l]; cv.n_mem = cnt_m[l + 1] - cnt_m[l];
      cv.log = cl.data() + cnt_l[l]; cv.n_log = cnt_l[l + 1] - cnt_l[l];
      cv.aux = ca.data() + cnt_a[l]; cv.n_aux = cnt_a[l + 1] - cnt_a[l];
      sink.cycle(cv);
    }
  }
}

struct MaterialiseSink {  // -> the per-instance arrays of zkw_instance_trace
  WaveTrace& wt;
  void cycle(const CycleView& cv) {
    const uint32_t l = cv.lane;
    wt.records[l].push_back(*cv.record);
    wt.mem[l].insert(wt.mem[l].end(), cv.mem, cv.mem + cv.n_mem);
    wt.log[l].insert(wt.log[l].end(), cv.log, cv.log + cv.n_log);
    wt.aux[l].insert(wt.aux[l].end(), cv.aux, cv.aux + cv.n_aux);
    wt.mem_off[l].push_back((uint32_t)wt.mem[l].size());
    wt.log_off[l].push_back((uint32_t)wt.log[l].size());
    wt.aux_off[l].push_back((uint32_t)wt.aux[l].size());
  }
};

static std::unique_ptr<WaveTrace> materialise_wave(const BatchInputs& in, uint32_t w, const WaveView& v, const uint32_t* ncyc) {
  const uint32_t L = v.L;
  auto wt = std::make_unique<WaveTrace>();
  wt->records.resize(L); wt->mem.resize(L); wt->log.resize(L); wt->aux.resize(L);
  wt->mem_off.resize(L); wt->log_off.resize(L); wt->aux_off.resize(L);
  for (uint32_t l = 0; l < L; l++) {
    wt->records[l].reserve(ncyc[l]);
    wt->mem_off[l].reserve(ncyc[l] + 1); wt->log_off[l].reserve(ncyc[l] + 1); wt->aux_off[l].reserve(ncyc[l] + 1);
    wt->mem_off[l].assign(1, 0); wt->log_off[l].assign(1, 0); wt->aux_off[l].assign(1, 0);
  }
  MaterialiseSink sink{*wt};
  WalkScratch scratch;
  walk_wave(scratch, in, w, v, ncyc, sink);
  return wt;
}

// zkw_instance_trace.final_state from the device's scalars, the lane's 30 register chunks (`stride` units apart) and its
// current callstack entry
static void fill_final_state(zkw_vm_local_state* out, const zkw_dev_scalars& sc, const uint4* regs, size_t stride, const zkw_dev_entry& cur) {
  zkw_vm_local_state& fs = *out;
  std::memcpy(fs.previous_code_word.l, sc.prev_code_word, 32);
  for (uint32_t ch = 0; ch < ZKW_REG_CHUNKS; ch++) std::memcpy((uint8_t*)fs.registers + 16 * ch, &regs[ch * stride], 16);
  fs.register_ptr_bitmap = (uint16_t)sc.ptr_bitmap;
  fs.flags = sc.flags & 7u;
  fs.pending_exception = (sc.flags >> 3) & 1u;
  fs.previous_code_memory_page = sc.prev_code_page;
  fs.timestamp = sc.timestamp;
  fs.monotonic_cycle_counter = sc.cycle_counter;
  fs.spent_pubdata_counter = sc.spent_pubdata;
  fs.memory_page_counter = sc.memory_page_counter;
  fs.absolute_execution_step = sc.absolute_execution_step;
  fs.current_ergs_per_pubdata_byte = sc.ergs_per_pubdata;
  fs.tx_number_in_block = (uint16_t)sc.tx_number;
  fs.previous_super_pc = (uint16_t)sc.prev_super_pc;
  fs.callstack_depth = sc.depth;
  std::memcpy(fs.context_u128_register, sc.ctx_u128_reg, 16);
  fs.current = cur.e;
}

// The link flags of a block: memory reads travel without their values when the host holds, for EVERY batch of the block, the heap
// images the step ran on (the shadow memory of walk_wave starts from them)
static uint32_t pack_flags(const zkw_ctx* c, zkw_batch* const* bs, uint32_t n) {
  if (c->opt_read_values) return 0;  // (the round-5 format: every page, every value)
  uint32_t flags = ZKW_PACK_NO_READ_VALUES | ZKW_PACK_IMPLIED_PAGES | ZKW_PACK_SLIM_TAILS | ZKW_PACK_SPARSE_DELTAS | ZKW_PACK_DELTA_TAILS;
  for (uint32_t i = 0; i < n; i++)
    if (!bs[i]->inputs || !bs[i]->inputs->heaps_known) flags &= ~ZKW_PACK_NO_READ_VALUES;
  return flags & ~c->opt_link_flags_off;
}

// The on-demand path of zkw_batch_get_instance_trace: ONE wave of a synced batch through the pack kernel into a pinned block
// of the batch (grown on demand), then the same rebuild as a delivered step — every parity test that reads a trace runs the
// pack kernel and the link-format rebuild.
static int pack_one_wave(zkw_batch* b, uint32_t w, uint32_t flags, std::unique_ptr<WaveTrace>& out) {
  zkw_ctx* c = b->ctx;
  const uint32_t L = b->L;
  std::vector<uint32_t> ncyc(L, 0);
  for (uint32_t l = 0; l < L; l++) {
    const uint32_t i = w * L + l;
    if (i < b->n) ncyc[l] = b->h_scalars[i].n_cycles;
  }
  // capacity: everything the wave's streams hold (the cursors are on the host since zkw_batch_sync)
  const uint32_t* hc = &b->h_cursors[(size_t)w * 4];
  const uint32_t n_mem = std::min(hc[0], b->cap_mem), n_log = std::min(hc[1], b->cap_log), n_aux = std::min(hc[2], b->cap_aux), n_delta = std::min(hc[3], b->cap_delta);
  const uint64_t fixed = ZKW_PACK_HEADER_UNITS + ZKW_PACK_BATCH_UNITS + ZKW_PACK_WAVE_UNITS;
  const uint64_t need = fixed + zkw_pack_wave_units_worst(b->lim.max_cycles, L, n_delta, n_mem, n_log, n_aux) + 4 + 16;
  if (need >= (1ull << 32)) {
    c->last_error = "wave trace beyond 64 GB";
    return ZKW_ERR_LIMIT;
  }
  if (b->h_pack_units < need) {
    if (b->h_pack) (void)hipHostFree(b->h_pack);
    b->h_pack = nullptr;
    HIP_TRY(c, hipHostMalloc((void**)&b->h_pack, (size_t)need * 16, hipHostMallocDefault));
    b->h_pack_units = need;
  }
  if (!b->d_pack_state.p) HIP_TRY(c, b->d_pack_state.alloc(4));
  const uint32_t state0[4] = {(uint32_t)fixed, 0, 0, 0};
  HIP_TRY(c, hipMemcpyAsync(b->d_pack_state.p, state0, sizeof state0, hipMemcpyHostToDevice, b->run_stream));
  zkw_pack_args A;
  std::memset(&A, 0, sizeof A);
  A.kp[0] = b->d_kp.p;
  A.wave_base[1] = b->n_waves;
  A.dst = b->h_pack; A.state = b->d_pack_state.p; A.dst_units = (uint32_t)need; A.n_batches = 1;
  A.wave_table = ZKW_PACK_HEADER_UNITS + ZKW_PACK_BATCH_UNITS; A.with_instances = 0; A.only_wave = w;
  A.flags = flags;
  HIP_TRY(c, zkw_launch_pack(&A, (uint32_t)c->wave_width, 1, b->run_stream));
  uint32_t state1[4] = {0, 0, 0, 0};
  HIP_TRY(c, hipMemcpyAsync(state1, b->d_pack_state.p, sizeof state1, hipMemcpyDeviceToHost, b->run_stream));
  HIP_TRY(c, hipStreamSynchronize(b->run_stream));
  zkw_pack_wave e;
  std::memcpy(&e, b->h_pack + A.wave_table, sizeof e);
  WaveView v;
  if (state1[1] != 0 || !wave_view(b->h_pack, b->h_pack_units, e, L, A.flags, v)) {
    c->last_error = "pack kernel: the wave did not fit its block";
    return ZKW_ERR_LIMIT;
  }
  out = materialise_wave(*b->inputs, w, v, ncyc.data());
  return ZKW_OK;
}

template <class T>
static bool same_rows(const std::vector<std::vector<T>>& a, const std::vector<std::vector<T>>& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i++)
    if (a[i].size() != b[i].size() || (!a[i].empty() && std::memcmp(a[i].data(), b[i].data(), a[i].size() * sizeof(T)) != 0)) return false;
  return true;
}

static int build_wave(zkw_batch* b, uint32_t w) {
  zkw_ctx* c = b->ctx;
  const uint32_t flags = pack_flags(c, &b, 1);
  std::unique_ptr<WaveTrace> wt;
  int rc = pack_one_wave(b, w, flags, wt);
  if (rc != ZKW_OK && !c->link_checked && c->opt_link_selfcheck && flags != 0) {  // (the format in use could not even be read back: the self-check below, failed)
    c->link_checked = true;
    rc = pack_one_wave(b, w, 0, wt);
    if (rc == ZKW_OK) {
      std::fprintf(stderr, "zkw: LINK FORMAT SELF-CHECK FAILED (flags %u): wave %u could not be rebuilt; this context keeps the plain format (zkw_delivered.link_flags = 0). Please report.\n", flags, w);
      c->opt_link_flags_off = 0xffffffffu;
    }
  }
  if (rc != ZKW_OK) return rc;
  // Self-check of the link format, once per context, on the first wave anybody reads: the same wave packed in the plain format
  // (every page, every value, 16-byte tails, 32-byte deltas: nothing for the rebuild to derive) must rebuild to the same trace.
  // If it does not — a device this format's scans were never run on — the context says so on stderr and keeps the plain format.
  if (!c->link_checked && c->opt_link_selfcheck && flags != 0) {
    c->link_checked = true;
    std::unique_ptr<WaveTrace> plain;
    const int rc0 = pack_one_wave(b, w, 0, plain);
    if (rc0 != ZKW_OK) return rc0;
    const bool same = c->opt_link_selfcheck != 2 /* (2: the test hook — behave as after a mismatch) */ && same_rows(wt->records, plain->records) && same_rows(wt->mem, plain->mem) &&
                      same_rows(wt->log, plain->log) && same_rows(wt->aux, plain->aux) && same_rows(wt->mem_off, plain->mem_off);
    if (!same) {
      std::fprintf(stderr, "zkw: LINK FORMAT SELF-CHECK FAILED (flags %u): wave %u rebuilds differently from the plain format; this context keeps the plain format "
                           "(zkw_delivered.link_flags = 0). Please report.\n", flags, w);
      c->opt_link_flags_off = 0xffffffffu;
      wt = std::move(plain);
    }
  }
  b->wave_cache[w] = std::move(wt);
  return ZKW_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// zkw_delivery: whole steps in a persistent pinned ring (include/zkw.h)
// ---------------------------------------------------------------------------------------------------------------------
struct DeliverySlot {
  uint4* h = nullptr;  // the pinned block ...
  uint4* d = nullptr;  // ... and its address on the device
  uint64_t units = 0;
  int state = 0;       // 0 free, 1 submitted, 2 landed (the host has waited for it)
  uint32_t ticket = 0;
  std::vector<std::shared_ptr<const BatchInputs>> inputs;  // per batch of the block: what ITS step ran on (the batches themselves may be
                                                           // restaged, uploaded again or destroyed while the ticket is read)
  hipEvent_t ev_run = nullptr, ev_k0 = nullptr, ev_k1 = nullptr, ev_done = nullptr;
  uint32_t* d_state = nullptr;          // [4] allocation cursor, overflow flag
  zkw_pack_batch* d_batches = nullptr;  // [ZKW_PACK_MAX] device copy of the block's batch table
  uint32_t wave_table = 0, n_waves = 0;
  std::map<std::pair<uint32_t, uint32_t>, std::unique_ptr<WaveTrace>> cache;  // (batch, wave) -> materialised traces
};

struct zkw_delivery {
  zkw_ctx* ctx = nullptr;
  std::vector<DeliverySlot> slots;
  hipStream_t stream = nullptr;
  uint32_t next_ticket = 0;
  uint32_t pack_blocks = 64;
  // the pool: persistent workers, one job at a time (a job = a function of the thread index, run once by every worker)
  uint32_t n_threads = 1;
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::function<void(uint32_t)> job;
  uint64_t job_gen = 0;
  uint32_t job_left = 0;
  bool stopping = false;
};

static void delivery_worker(zkw_delivery* d, uint32_t t) {
  uint64_t seen = 0;
  for (;;) {
    std::function<void(uint32_t)> fn;
    {
      std::unique_lock<std::mutex> lk(d->mu);
      d->cv_work.wait(lk, [&] { return d->stopping || d->job_gen != seen; });
      if (d->stopping) return;
      seen = d->job_gen;
      fn = d->job;
    }
    fn(t);
    {
      std::lock_guard<std::mutex> lk(d->mu);
      if (--d->job_left == 0) d->cv_done.notify_all();
    }
  }
}
static void delivery_run(zkw_delivery* d, std::function<void(uint32_t)> fn) {
  if (d->workers.empty()) {  // (one thread: the caller's)
    fn(0);
    return;
  }
  std::unique_lock<std::mutex> lk(d->mu);
  d->job = std::move(fn);
  d->job_left = (uint32_t)d->workers.size();
  d->job_gen++;
  d->cv_work.notify_all();
  d->cv_done.wait(lk, [&] { return d->job_left == 0; });
}

struct FoldSink {  // the built-in consumer of zkw_delivery_replay: reads every byte it is handed
  uint64_t acc = 0, cycles = 0;
  // sum_j (u64[j] ^ K (j + 1 + w0)), K = 2^64 / phi: position-sensitive, and a plain sum over the records — the order in which
  // threads get to the waves does not matter.  An XOR and an add per word (two vector operations per 16 bytes even with the
  // baseline x86-64 instruction set this file is compiled for): the stand-in tracer should not cost more than the walk.
  static inline __attribute__((always_inline)) uint64_t fold_body(const void* p, uint32_t words, uint64_t w0) {
    const uint64_t* u = (const uint64_t*)p;
    uint64_t a = 0, k = 0x9E3779B97F4A7C15ull * (w0 + 1);
    for (uint32_t j = 0; j < words; j++, k += 0x9E3779B97F4A7C15ull) a += u[j] ^ k;
    return a;
  }
  static uint64_t fold_base(const void* p, uint32_t words, uint64_t w0) { return fold_body(p, words, w0); }
  // (the same loop compiled for 256-bit integer vectors where the host has them: the file itself is built for the baseline instruction set)
  __attribute__((target("avx2"))) static uint64_t fold_avx2(const void* p, uint32_t words, uint64_t w0) { return fold_body(p, words, w0); }
  static uint64_t fold(const void* p, uint32_t words, uint64_t w0) {
    static const bool wide = __builtin_cpu_supports("avx2") != 0;
    return wide ? fold_avx2(p, words, w0) : fold_base(p, words, w0);
  }
  void cycle(const CycleView& cv) {
    acc += fold(cv.record, 64, 1);
    for (uint32_t i = 0; i < cv.n_mem; i++) acc += fold(cv.mem + i, 6, 3);
    for (uint32_t i = 0; i < cv.n_log; i++) acc += fold(cv.log + i, 16, 5);
    for (uint32_t i = 0; i < cv.n_aux; i++) acc += fold(cv.aux + i, 32, 7);
    cycles++;
  }
};
struct CallbackSink {
  zkw_cycle_fn fn;
  void* user;
  uint32_t thread, batch_index, first_instance;
  uint64_t cycles = 0;
  void cycle(const CycleView& cv) {
    fn(user, thread, batch_index, first_instance + cv.lane, cv.cycle, cv.record, cv.mem, cv.n_mem, cv.log, cv.n_log, cv.aux, cv.n_aux);
    cycles++;
  }
};

static DeliverySlot* delivery_slot(zkw_delivery* d, uint32_t ticket, bool landed) {
  if (!d || d->slots.empty()) return nullptr;
  DeliverySlot& sl = d->slots[ticket % d->slots.size()];
  if (sl.state == 0 || sl.ticket != ticket || (landed && sl.state != 2)) return nullptr;
  return &sl;
}
// the view of wave (bi, w) of a landed slot + the cycle counts of its lanes (from the packed scalars)
static bool delivery_wave(const DeliverySlot& sl, uint32_t bi, uint32_t w, WaveView& v, uint32_t* ncyc) {
  const zkw_pack_batch* pbs = (const zkw_pack_batch*)(sl.h + ZKW_PACK_HEADER_UNITS);
  const zkw_pack_batch& pb = pbs[bi];
  const zkw_pack_wave* wt = (const zkw_pack_wave*)(sl.h + sl.wave_table);
  if (!wave_view(sl.h, sl.units, wt[pb.first_wave + w], pb.L, ((const zkw_pack_header*)sl.h)->flags, v)) return false;
  const zkw_dev_scalars* sc = (const zkw_dev_scalars*)(sl.h + pb.scalars_off);
  for (uint32_t l = 0; l < pb.L; l++) {
    const uint32_t i = w * pb.L + l;
    ncyc[l] = i < pb.n_instances ? sc[i].n_cycles : 0u;
  }
  return true;
}

extern "C" {

int zkw_batch_get_instance_trace(zkw_batch* b, uint32_t instance, zkw_instance_trace* out) {
  if (!b || !out || instance >= b->n) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->synced) {
    int rc = zkw_batch_sync(b);
    if (rc != ZKW_OK) return rc;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t w = instance / b->L, l = instance % b->L;
  if (!b->wave_cache.count(w)) {
    int rc = build_wave(b, w);
    if (rc != ZKW_OK) return rc;
  }
  WaveTrace& wt = *b->wave_cache[w];
  const zkw_dev_scalars& sc = b->h_scalars[instance];
  std::memset(out, 0, sizeof *out);
  out->status = sc.status;
  out->n_cycles = sc.n_cycles;
  out->n_mem = (uint32_t)wt.mem[l].size();
  out->n_log = (uint32_t)wt.log[l].size();
  out->n_aux = (uint32_t)wt.aux[l].size();
  out->records = wt.records[l].data();
  out->mem = wt.mem[l].data();
  out->log = wt.log[l].data();
  out->aux = wt.aux[l].data();
  out->mem_off = wt.mem_off[l].data();
  out->log_off = wt.log_off[l].data();
  out->aux_off = wt.aux_off[l].data();
  // final VmLocalState: scalars + register file + current callstack entry
  std::vector<uint4> regs(ZKW_REG_CHUNKS);
  for (uint32_t ch = 0; ch < ZKW_REG_CHUNKS; ch++)
    HIP_TRY(c, hipMemcpy(&regs[ch], b->d_regs.p + ((size_t)w * ZKW_REG_CHUNKS + ch) * b->L + l, sizeof(uint4), hipMemcpyDeviceToHost));
  zkw_dev_entry cur;
  HIP_TRY(c, hipMemcpy(&cur, b->d_callstack.p + (size_t)instance * (b->lim.max_callstack_depth + 1) + sc.depth, sizeof cur, hipMemcpyDeviceToHost));
  fill_final_state(&out->final_state, sc, regs.data(), 1, cur);
  return ZKW_OK;
}

int zkw_delivery_slot_bytes(zkw_batch* const* batches, uint32_t n_batches, uint64_t* worst_case) {
  if (!batches || !n_batches || !worst_case) return ZKW_ERR_INVALID;
  uint64_t units = ZKW_PACK_HEADER_UNITS + (uint64_t)n_batches * ZKW_PACK_BATCH_UNITS;
  for (uint32_t i = 0; i < n_batches; i++) {
    const zkw_batch* b = batches[i];
    if (!b || !b->uploaded) return ZKW_ERR_INVALID;
    units += (uint64_t)b->n_waves * ZKW_PACK_WAVE_UNITS + (uint64_t)b->n * 16 + (uint64_t)b->n_waves * ZKW_REG_CHUNKS * b->L;
    units += (uint64_t)b->n_waves * (zkw_pack_wave_units_worst(b->lim.max_cycles, b->L, b->cap_delta, b->cap_mem, b->cap_log, b->cap_aux) + 4);
  }
  *worst_case = units * 16;
  return ZKW_OK;
}

int zkw_delivery_create(zkw_ctx* c, uint32_t n_slots, uint64_t slot_bytes, uint32_t host_threads, zkw_delivery** out) {
  if (!c || !out || n_slots == 0 || slot_bytes < 4096 || host_threads == 0 || host_threads > 4096) return ZKW_ERR_INVALID;
  if (slot_bytes / 16 >= (1ull << 32)) {
    c->last_error = "zkw_delivery_create: a slot holds at most 64 GB (32-bit offsets in 16-byte units)";
    return ZKW_ERR_LIMIT;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  auto* d = new zkw_delivery();
  d->ctx = c;
  d->n_threads = host_threads;
  if (c->opt_pack_blocks) d->pack_blocks = c->opt_pack_blocks;
  d->slots.resize(n_slots);
  auto fail = [&](hipError_t e, const char* what) {
    c->last_error = std::string(what) + ": " + hipGetErrorString(e);
    zkw_delivery_destroy(d);
    return ZKW_ERR_DEVICE;
  };
  hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return fail(e, "hipStreamCreateWithFlags");
  for (DeliverySlot& sl : d->slots) {
    sl.units = slot_bytes / 16;
    if ((e = hipHostMalloc((void**)&sl.h, sl.units * 16, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc (the pinned ring)");
    if ((e = hipHostGetDevicePointer((void**)&sl.d, sl.h, 0)) != hipSuccess) return fail(e, "hipHostGetDevicePointer");
    if ((e = hipMalloc((void**)&sl.d_state, 16)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMalloc((void**)&sl.d_batches, sizeof(zkw_pack_batch) * ZKW_PACK_MAX)) != hipSuccess) return fail(e, "hipMalloc");
    for (hipEvent_t* ev : {&sl.ev_run, &sl.ev_k0, &sl.ev_k1})
      if ((e = hipEventCreate(ev)) != hipSuccess) return fail(e, "hipEventCreate");
    // (the event a host thread waits on: blocking, so that the waiter sleeps — the host's cores belong to the replay)
    if ((e = hipEventCreateWithFlags(&sl.ev_done, hipEventBlockingSync)) != hipSuccess) return fail(e, "hipEventCreateWithFlags");
  }
  if (host_threads > 1)
    for (uint32_t t = 0; t < host_threads; t++) d->workers.emplace_back(delivery_worker, d, t);
  *out = d;
  return ZKW_OK;
}

void zkw_delivery_destroy(zkw_delivery* d) {
  if (!d) return;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    d->stopping = true;
  }
  d->cv_work.notify_all();
  for (std::thread& t : d->workers) t.join();
  (void)hipSetDevice(d->ctx->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  for (DeliverySlot& sl : d->slots) {
    if (sl.h) (void)hipHostFree(sl.h);
    if (sl.d_state) (void)hipFree(sl.d_state);
    if (sl.d_batches) (void)hipFree(sl.d_batches);
    for (hipEvent_t ev : {sl.ev_run, sl.ev_k0, sl.ev_k1, sl.ev_done})
      if (ev) (void)hipEventDestroy(ev);
  }
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int zkw_delivery_submit(zkw_delivery* d, zkw_batch* const* batches, uint32_t n, void* run_stream, uint32_t* ticket) {
  if (!d || !ticket) return ZKW_ERR_INVALID;
  int rc = check_group(batches, n);
  if (rc != ZKW_OK) return rc;
  zkw_ctx* c = d->ctx;
  if (batches[0]->ctx != c) return ZKW_ERR_INVALID;
  DeliverySlot& sl = d->slots[d->next_ticket % d->slots.size()];
  if (sl.state != 0) {
    c->last_error = "zkw_delivery_submit: the ring is full (release the oldest ticket)";
    return ZKW_ERR_LIMIT;
  }
  for (uint32_t i = 0; i < n; i++)
    if (!batches[i]->ran) {
      c->last_error = "zkw_delivery_submit: a batch has not run";
      return ZKW_ERR_NOT_RUN;
    }
  HIP_TRY(c, hipSetDevice(c->device));
  // the fixed part of the block: header, batch table, wave table, the instance sections — written here, by the host
  zkw_pack_args A;
  std::memset(&A, 0, sizeof A);
  uint64_t at = ZKW_PACK_HEADER_UNITS + (uint64_t)n * ZKW_PACK_BATCH_UNITS;
  uint32_t waves = 0;
  for (uint32_t i = 0; i < n; i++) waves += batches[i]->n_waves;
  const uint64_t wave_table = at;
  at += (uint64_t)waves * ZKW_PACK_WAVE_UNITS;
  zkw_pack_batch* pbs = (zkw_pack_batch*)(sl.h + ZKW_PACK_HEADER_UNITS);
  uint32_t first_wave = 0;
  for (uint32_t i = 0; i < n; i++) {
    const zkw_batch* b = batches[i];
    zkw_pack_batch pb;
    pb.n_instances = b->n; pb.L = b->L; pb.n_waves = b->n_waves; pb.first_wave = first_wave; pb.max_cycles = b->lim.max_cycles;
    pb.scalars_off = (uint32_t)at; at += (uint64_t)b->n * 8;
    pb.regs_off = (uint32_t)at; at += (uint64_t)b->n_waves * ZKW_REG_CHUNKS * b->L;
    pb.entries_off = (uint32_t)at; at += (uint64_t)b->n * 8;
    if (at >= sl.units) {
      c->last_error = "zkw_delivery_submit: the slot is smaller than the fixed part of this step";
      return ZKW_ERR_LIMIT;
    }
    pbs[i] = pb;
    A.kp[i] = b->d_kp.p;
    A.wave_base[i + 1] = A.wave_base[i] + b->n_waves;
    first_wave += b->n_waves;
  }
  zkw_pack_header* hd = (zkw_pack_header*)sl.h;
  std::memset(hd, 0, sizeof *hd);
  hd->magic = ZKW_PACK_MAGIC; hd->version = ZKW_PACK_VERSION; hd->n_batches = n; hd->n_waves = waves; hd->fixed_units = (uint32_t)at; hd->with_instances = 1;
  hd->used_units = (uint32_t)at;  // the allocation cursor starts behind the fixed part (copied to the device below, and back behind the kernel)
  hd->overflow = 0;
  hd->flags = A.flags = pack_flags(c, batches, n);
  A.batches = sl.d_batches; A.dst = sl.d; A.state = sl.d_state; A.dst_units = (uint32_t)sl.units; A.n_batches = n;
  A.wave_table = (uint32_t)wave_table; A.with_instances = 1; A.only_wave = 0xffffffffu;
  hipStream_t rs = (hipStream_t)run_stream;
  HIP_TRY(c, hipEventRecord(sl.ev_run, rs));
  HIP_TRY(c, hipStreamWaitEvent(d->stream, sl.ev_run, 0));
  HIP_TRY(c, hipMemcpyAsync(sl.d_state, &hd->used_units, 8, hipMemcpyHostToDevice, d->stream));  // (from the pinned slot itself)
  HIP_TRY(c, hipMemcpyAsync(sl.d_batches, pbs, sizeof(zkw_pack_batch) * n, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(c, hipEventRecord(sl.ev_k0, d->stream));
  HIP_TRY(c, zkw_launch_pack(&A, (uint32_t)c->wave_width, d->pack_blocks, d->stream));
  HIP_TRY(c, hipEventRecord(sl.ev_k1, d->stream));
  HIP_TRY(c, hipMemcpyAsync(&hd->used_units, sl.d_state, 8, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(c, hipEventRecord(sl.ev_done, d->stream));
  sl.state = 1;
  sl.ticket = d->next_ticket;
  sl.inputs.clear();
  for (uint32_t i = 0; i < n; i++) sl.inputs.push_back(batches[i]->inputs);
  sl.wave_table = (uint32_t)wave_table;
  sl.n_waves = waves;
  sl.cache.clear();
  *ticket = d->next_ticket++;
  return ZKW_OK;
}

int zkw_delivery_order_after(zkw_delivery* d, uint32_t ticket, void* hip_stream) {
  DeliverySlot* sl = delivery_slot(d, ticket, false);
  if (!sl) return ZKW_ERR_INVALID;
  HIP_TRY(d->ctx, hipSetDevice(d->ctx->device));
  HIP_TRY(d->ctx, hipStreamWaitEvent((hipStream_t)hip_stream, sl->ev_done, 0));
  return ZKW_OK;
}

int zkw_delivery_wait(zkw_delivery* d, uint32_t ticket, zkw_delivered* info) {
  DeliverySlot* sl = delivery_slot(d, ticket, false);
  if (!sl) return ZKW_ERR_INVALID;
  zkw_ctx* c = d->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventSynchronize(sl->ev_done));
  sl->state = 2;
  const zkw_pack_header* hd = (const zkw_pack_header*)sl->h;
  if (info) {
    float ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms, sl->ev_k0, sl->ev_k1));
    info->bytes = (uint64_t)std::min<uint64_t>(hd->used_units, sl->units) * 16;
    info->pack_ms = ms;
    info->n_batches = hd->n_batches; info->n_waves = hd->n_waves; info->overflow = hd->overflow; info->link_flags = hd->flags;
  }
  if (hd->overflow) {
    c->last_error = "zkw_delivery_wait: the step did not fit its slot (slot_bytes too small)";
    return ZKW_ERR_LIMIT;
  }
  return ZKW_OK;
}

int zkw_delivery_get_instance_trace(zkw_delivery* d, uint32_t ticket, uint32_t bi, uint32_t instance, zkw_instance_trace* out) {
  DeliverySlot* sl = delivery_slot(d, ticket, true);
  if (!sl || !out || bi >= sl->inputs.size()) return ZKW_ERR_INVALID;
  const zkw_pack_batch& pb = ((const zkw_pack_batch*)(sl->h + ZKW_PACK_HEADER_UNITS))[bi];
  if (instance >= pb.n_instances) return ZKW_ERR_INVALID;
  const uint32_t L = pb.L, w = instance / L, l = instance % L;
  auto key = std::make_pair(bi, w);
  if (!sl->cache.count(key)) {
    WaveView v;
    std::vector<uint32_t> ncyc(L, 0);
    if (!delivery_wave(*sl, bi, w, v, ncyc.data())) {
      d->ctx->last_error = "zkw_delivery_get_instance_trace: the wave was not delivered (overflow)";
      return ZKW_ERR_LIMIT;
    }
    sl->cache[key] = materialise_wave(*sl->inputs[bi], w, v, ncyc.data());
  }
  WaveTrace& wt = *sl->cache[key];
  const zkw_dev_scalars& sc = ((const zkw_dev_scalars*)(sl->h + pb.scalars_off))[instance];
  std::memset(out, 0, sizeof *out);
  out->status = sc.status;
  out->n_cycles = sc.n_cycles;
  out->n_mem = (uint32_t)wt.mem[l].size(); out->n_log = (uint32_t)wt.log[l].size(); out->n_aux = (uint32_t)wt.aux[l].size();
  out->records = wt.records[l].data(); out->mem = wt.mem[l].data(); out->log = wt.log[l].data(); out->aux = wt.aux[l].data();
  out->mem_off = wt.mem_off[l].data(); out->log_off = wt.log_off[l].data(); out->aux_off = wt.aux_off[l].data();
  const uint4* regs = sl->h + pb.regs_off + (size_t)w * ZKW_REG_CHUNKS * L + l;
  fill_final_state(&out->final_state, sc, regs, L, ((const zkw_dev_entry*)(sl->h + pb.entries_off))[instance]);
  return ZKW_OK;
}

int zkw_delivery_replay(zkw_delivery* d, uint32_t ticket, zkw_cycle_fn fn, void* user, uint64_t* n_cycles, uint64_t* checksum) {
  DeliverySlot* sl = delivery_slot(d, ticket, true);
  if (!sl) return ZKW_ERR_INVALID;
  const zkw_pack_batch* pbs = (const zkw_pack_batch*)(sl->h + ZKW_PACK_HEADER_UNITS);
  const uint32_t nb = (uint32_t)sl->inputs.size();
  std::atomic<uint32_t> next{0};
  std::atomic<uint64_t> cycles{0}, acc{0};
  std::atomic<uint32_t> missing{0};
  delivery_run(d, [&](uint32_t t) {
    uint64_t my_cycles = 0, my_acc = 0;
    std::vector<uint32_t> ncyc(ZKW_WAVE, 0);
    WalkScratch scratch;
    for (;;) {
      const uint32_t gw = next.fetch_add(1);
      if (gw >= sl->n_waves) break;
      uint32_t bi = 0;
      while (bi + 1 < nb && pbs[bi + 1].first_wave <= gw) bi++;
      const uint32_t w = gw - pbs[bi].first_wave;
      WaveView v;
      if (!delivery_wave(*sl, bi, w, v, ncyc.data())) {
        missing.fetch_add(1);
        continue;
      }
      if (fn) {
        CallbackSink sink{fn, user, t, bi, w * pbs[bi].L};
        walk_wave(scratch, *sl->inputs[bi], w, v, ncyc.data(), sink);
        my_cycles += sink.cycles;
      } else {
        FoldSink sink;
        walk_wave(scratch, *sl->inputs[bi], w, v, ncyc.data(), sink);
        my_cycles += sink.cycles;
        my_acc += sink.acc;
      }
    }
    cycles.fetch_add(my_cycles);
    acc.fetch_add(my_acc);
  });
  if (n_cycles) *n_cycles = cycles.load();
  if (checksum) *checksum = acc.load();
  if (missing.load()) {
    d->ctx->last_error = "zkw_delivery_replay: waves were not delivered (overflow)";
    return ZKW_ERR_LIMIT;
  }
  return ZKW_OK;
}

int zkw_delivery_release(zkw_delivery* d, uint32_t ticket) {
  DeliverySlot* sl = delivery_slot(d, ticket, false);
  if (!sl) return ZKW_ERR_INVALID;
  if (sl->state == 1) {  // not waited for: the block may still be in flight
    HIP_TRY(d->ctx, hipSetDevice(d->ctx->device));
    HIP_TRY(d->ctx, hipEventSynchronize(sl->ev_done));
  }
  sl->state = 0;
  sl->cache.clear();
  sl->inputs.clear();
  return ZKW_OK;
}

// Pinned staging of a batch's fresh inputs ([states | heap images], instance-major as the C ABI takes them) + the device buffer the
// H2D copies land in.  The host side is a small ring of buffers: a buffer whose heap images a BatchInputs still reads (the current
// staging of the batch, or a delivered step whose ticket is held) is not handed out again — the shadow memory of the rebuild reads
// the images right there, nothing is copied to keep them.  acquire_stage picks the buffer the next restage writes into.
static int acquire_stage(zkw_batch* b) {
  zkw_ctx* c = b->ctx;
  const size_t bytes = (size_t)b->n * sizeof(zkw_vm_local_state) + (size_t)b->n * b->heap_image_words * 32 + 16;
  const uint32_t cap = c->opt_staging_buffers ? c->opt_staging_buffers : 4u;
  if (b->d_stage_bytes < bytes) {
    b->d_stage.release();
    HIP_TRY(c, b->d_stage.alloc(bytes));
    b->d_stage_bytes = bytes;
  }
  uint32_t pick = 0xffffffffu;
  const uint32_t N = (uint32_t)b->stage.size();
  for (uint32_t k = 0; k < N && pick == 0xffffffffu; k++) {
    const uint32_t j = (b->stage_cur + k) % N;
    if (!b->stage[j].mem || b->stage[j].mem.use_count() == 1) pick = j;  // nobody but the batch holds it
  }
  if (pick == 0xffffffffu) {
    if (N >= cap) {
      c->last_error = "zkw_batch_restage / zkw_batch_staging: all " + std::to_string(N) + " staging buffers of the batch still hold the heap images of steps whose delivery "
                      "tickets are held — release tickets first (or raise ZKW_OPT_STAGING_BUFFERS)";
      return ZKW_ERR_LIMIT;
    }
    b->stage.emplace_back();
    pick = N;
  }
  StageBuf& sb = b->stage[pick];
  if (sb.busy) {  // the copies of the restage that used it last still read it
    HIP_TRY(c, hipEventSynchronize(sb.ev));
    sb.busy = false;
  }
  if (!sb.mem || sb.mem->bytes < bytes) {
    auto m = std::make_shared<StageMem>();
    HIP_TRY(c, hipHostMalloc((void**)&m->h, bytes, hipHostMallocDefault));
    m->bytes = bytes;
    sb.mem = m;
  }
  if (!sb.ev) HIP_TRY(c, hipEventCreate(&sb.ev));
  b->stage_cur = pick;
  return ZKW_OK;
}

int zkw_batch_staging(zkw_batch* b, zkw_vm_local_state** states, zkw_u256** heap_words, uint32_t* n_heap_words) {
  if (!b || !states) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->uploaded) {
    c->last_error = "zkw_batch_staging: the batch has not been uploaded";
    return ZKW_ERR_INVALID;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = acquire_stage(b);
  if (rc != ZKW_OK) return rc;
  uint8_t* h = b->stage[b->stage_cur].mem->h;
  *states = (zkw_vm_local_state*)h;
  if (heap_words) *heap_words = (zkw_u256*)(h + (((size_t)b->n * sizeof(zkw_vm_local_state) + 15) & ~(size_t)15));
  if (n_heap_words) *n_heap_words = b->heap_image_words;
  return ZKW_OK;
}

int zkw_batch_restage(zkw_batch* b, const zkw_vm_local_state* states, const zkw_u256* heap_words, uint32_t n_heap_words, void* hip_stream) {
  if (!b || !states) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->uploaded) {
    c->last_error = "zkw_batch_restage: the batch has not been uploaded";
    return ZKW_ERR_INVALID;
  }
  const uint32_t n = b->n, himg = b->heap_image_words;
  if (heap_words && n_heap_words != himg) {
    c->last_error = "zkw_batch_restage: the heap images must be as long as the uploaded ones (" + std::to_string(himg) + " words)";
    return ZKW_ERR_INVALID;
  }
  // geometry is fixed at upload: callstack depth, code page and base page of the current frame (their blob and arena slot stay)
  for (uint32_t i = 0; i < n; i++) {
    const zkw_vm_local_state& o = b->staged[i].state;
    if (states[i].callstack_depth != o.callstack_depth || states[i].current.code_page != o.current.code_page || states[i].current.base_memory_page != o.current.base_memory_page ||
        states[i].current.is_local_frame != o.current.is_local_frame) {
      c->last_error = "zkw_batch_restage: instance " + std::to_string(i) + " changes its callstack depth / code page / base page (fixed at upload: upload again)";
      return ZKW_ERR_INVALID;
    }
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t st_bytes = (size_t)n * sizeof(zkw_vm_local_state), heap_off = (st_bytes + 15) & ~(size_t)15, heap_bytes = (size_t)n * himg * 32;
  // the caller filled the staging buffer zkw_batch_staging handed out: nothing to copy
  const bool in_place = !b->stage.empty() && b->stage[b->stage_cur].mem && (const uint8_t*)states == b->stage[b->stage_cur].mem->h;
  if (!in_place) {
    int rc = acquire_stage(b);  // a buffer nobody reads any more
    if (rc != ZKW_OK) return rc;
    std::memcpy(b->stage[b->stage_cur].mem->h, states, st_bytes);
  } else {
    // The caller wrote into the buffer zkw_batch_staging handed out.  If it kept the pointers from an EARLIER call and a delivered
    // step still reads its heap images out of this very buffer, they have just been overwritten under it: refuse instead of
    // rebuilding that step's memory reads from the wrong images.  (Holders: the batch, and its own current inputs.)
    const std::shared_ptr<StageMem>& m = b->stage[b->stage_cur].mem;
    const bool mine = b->inputs && b->inputs->heap_owner.get() == (const void*)m.get();
    const long expected = 1 + (mine ? 1 : 0);
    if (m.use_count() > expected || (mine && b->inputs.use_count() > 1)) {  // (a ticket shares the batch's inputs OBJECT, not the buffer)
      c->last_error = "zkw_batch_restage: the staging buffer passed in still holds the heap images of a delivered step whose ticket is held: "
                      "call zkw_batch_staging again for every restage (it hands out a buffer nobody reads)";
      return ZKW_ERR_INVALID;
    }
  }
  StageBuf& sb = b->stage[b->stage_cur];
  uint8_t* const h_stage = sb.mem->h;
  if (heap_words && (const uint8_t*)heap_words != h_stage + heap_off) std::memcpy(h_stage + heap_off, heap_words, heap_bytes);
  // the library's own copy of the initial states (what a trace is rebuilt onto); the staged heap vectors are not kept in step —
  // a later zkw_batch_upload needs zkw_batch_set_heap again
  for (uint32_t i = 0; i < n; i++) b->staged[i].state = states[i];
  {  // a NEW inputs object: the tickets of earlier steps keep the one their step ran on
    auto in = std::make_shared<BatchInputs>();
    in->states.assign(states, states + n);
    in->code = b->inputs->code;
    in->heap_words = himg;
    if (!heap_words) {  // the images stay what they were
      in->heap_data = b->inputs->heap_data;
      in->heap_owner = b->inputs->heap_owner;
      in->heaps_known = b->inputs->heaps_known;
    } else if ((c->opt_staging_buffers ? c->opt_staging_buffers : 4u) > 1) {
      // the images stay where they were handed over: the inputs hold the staging buffer (no copy), the batch takes another one
      // of its ring for the next restage
      in->heap_data = (const zkw_u256*)(h_stage + heap_off);
      in->heap_owner = sb.mem;
      in->heaps_known = true;
    }  // (a ring of ONE buffer is overwritten by the next restage: the images are then not kept, memory reads travel with values)
    b->inputs = in;
  }
  if (heap_words) b->heaps_restaged = true;
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(c, hipMemcpyAsync(b->d_stage.p, h_stage, st_bytes, hipMemcpyHostToDevice, st));
  if (heap_words && heap_bytes) HIP_TRY(c, hipMemcpyAsync(b->d_stage.p + heap_off, h_stage + heap_off, heap_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipEventRecord(sb.ev, st));
  sb.busy = true;
  zkw_restage_params R;
  std::memset(&R, 0, sizeof R);
  R.states = (const zkw_vm_local_state*)b->d_stage.p;
  R.heaps = heap_words ? (const uint4*)(b->d_stage.p + heap_off) : nullptr;
  R.regs0 = b->d_regs0.p; R.scalars0 = b->d_scalars0.p; R.callstack0 = b->d_callstack0.p; R.heap0 = b->d_heap0.p;
  R.n_instances = n; R.L = b->L; R.n_waves = b->n_waves; R.D = b->lim.max_callstack_depth; R.image_words = himg;
  R.F = b->lim.max_far_frames; R.frames0 = b->d_frames0.p;
  HIP_TRY(c, zkw_launch_restage(&R, (uint32_t)c->wave_width, st));
  b->full_reset_pending = true;  // the whole heap image goes into the arena, not just the words a run had dirtied
  zkw_batch* one[1] = {b};
  return enqueue_reset(one, 1, st);
}

// the two layouts of the expanded records: instance-major (stride_k == 1: the records of an instance are contiguous, the
// layout of zkw_instance_trace.records) or cycle-major (stride_i == 1: the records of a cycle are contiguous — what the kernel
// writes fastest: a wave's 64 records of a cycle are one 32 KB run)
static int expand_strides(zkw_ctx* c, uint64_t& si, uint64_t& sk, uint32_t max_cycles, uint32_t cycles_run, uint32_t count) {
  if (si == 0 && sk == 0) { si = max_cycles; sk = 1; }
  const bool instance_major = sk == 1 && si >= cycles_run, cycle_major = si == 1 && sk >= count;
  if (!instance_major && !cycle_major) {
    c->last_error = "expand_records: strides must be (>= cycles run, 1) — instance-major — or (1, >= instances) — cycle-major";
    return ZKW_ERR_INVALID;
  }
  return ZKW_OK;
}

int zkw_batch_expand_records(zkw_batch* b, uint32_t first, uint32_t count, void* dst_device, uint64_t instance_stride, uint64_t cycle_stride, void* hip_stream) {
  if (!b || !dst_device) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->uploaded) return ZKW_ERR_INVALID;
  if (!b->ran) return ZKW_ERR_NOT_RUN;
  if (count == 0 || first >= b->n || count > b->n - first) {
    c->last_error = "zkw_batch_expand_records: instance range outside the batch";
    return ZKW_ERR_INVALID;
  }
  int rc = expand_strides(c, instance_stride, cycle_stride, b->lim.max_cycles, b->cycles_run, count);
  if (rc != ZKW_OK) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const zkw_kparams* kp = b->d_kp.p;
  const uint32_t nw = b->n_waves;
  HIP_TRY(c, zkw_launch_expand(&kp, &dst_device, &nw, 1, instance_stride, cycle_stride, first, count, b->L, b->kp.wave_threads, b->cycles_run, (uint32_t)c->n_cus, (hipStream_t)hip_stream));
  return ZKW_OK;
}

int zkw_batches_expand_records(zkw_batch* const* batches, uint32_t n_batches, void* const* dst_device, uint64_t instance_stride, uint64_t cycle_stride, void* hip_stream) {
  int rc = check_group(batches, n_batches);
  if (rc != ZKW_OK) return rc;
  if (!dst_device) return ZKW_ERR_INVALID;
  zkw_ctx* c = batches[0]->ctx;
  uint32_t run = 0;
  for (uint32_t i = 0; i < n_batches; i++) {
    if (!batches[i]->ran) return ZKW_ERR_NOT_RUN;
    if (!dst_device[i]) return ZKW_ERR_INVALID;
    uint64_t si = instance_stride, sk = cycle_stride;
    rc = expand_strides(c, si, sk, batches[i]->lim.max_cycles, batches[i]->cycles_run, batches[i]->n);
    if (rc != ZKW_OK) return rc;
    if (instance_stride == 0 && cycle_stride == 0 && batches[i]->lim.max_cycles != batches[0]->lim.max_cycles) {
      c->last_error = "zkw_batches_expand_records: default strides need batches that share limits.max_cycles";
      return ZKW_ERR_INVALID;
    }
    run = std::max(run, batches[i]->cycles_run);
  }
  {
    uint64_t si = instance_stride, sk = cycle_stride;
    expand_strides(c, si, sk, batches[0]->lim.max_cycles, 0, 0);
    instance_stride = si; cycle_stride = sk;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  for (uint32_t at = 0; at < n_batches; at += 128) {  // (the by-value table of a launch holds 128 batches)
    const uint32_t n = std::min<uint32_t>(128, n_batches - at);
    const zkw_kparams* kp[128];
    void* dst[128];
    uint32_t nw[128];
    for (uint32_t i = 0; i < n; i++) {
      kp[i] = batches[at + i]->d_kp.p;
      dst[i] = dst_device[at + i];
      nw[i] = batches[at + i]->n_waves;
    }
    HIP_TRY(c, zkw_launch_expand(kp, dst, nw, n, instance_stride, cycle_stride, 0, 0xffffffffu, batches[at]->L, batches[at]->kp.wave_threads, run, (uint32_t)c->n_cus, (hipStream_t)hip_stream));
  }
  return ZKW_OK;
}

int zkw_batch_commit(zkw_batch* b, uint32_t queue_mask, void* hip_stream) {
  if (!b) return ZKW_ERR_INVALID;
  if (!b->uploaded) return ZKW_ERR_INVALID;
  return enqueue_commit(&b, 1, queue_mask, (hipStream_t)hip_stream);
}

int zkw_batch_net_states(zkw_batch* b, void* hip_stream) {
  if (!b) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->uploaded) return ZKW_ERR_INVALID;
  if (!b->ran) return ZKW_ERR_NOT_RUN;
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t per_log = b->lim.max_log_queries, per_aux = b->lim.max_aux_events, mark_cap = b->lim.max_callstack_depth + 2;
  if (!b->d_ns_params.p) {  // first use: buffers + parameter blocks (constant until the next upload)
    HIP_TRY(c, b->d_ns_log_idx.alloc((size_t)b->n * per_log));
    HIP_TRY(c, b->d_ns_log_cnt.alloc(b->n));
    HIP_TRY(c, b->d_ns_aux_idx.alloc((size_t)b->n * per_aux));
    HIP_TRY(c, b->d_ns_aux_cnt.alloc(b->n));
    HIP_TRY(c, b->d_ns_st_hist.alloc((size_t)b->n * 2 * per_log));
    HIP_TRY(c, b->d_ns_ev_hist.alloc((size_t)b->n * 2 * per_log));
    HIP_TRY(c, b->d_ns_rb_st.alloc((size_t)b->n * per_log));
    HIP_TRY(c, b->d_ns_rb_ev.alloc((size_t)b->n * per_log));
    HIP_TRY(c, b->d_ns_marks.alloc((size_t)b->n * mark_cap * 2));
    HIP_TRY(c, b->d_ns_counts.alloc((size_t)b->n * 4));
    zkw_commit_params CP[2];
    std::memset(CP, 0, sizeof CP);
    for (int k = 0; k < 2; k++) {
      zkw_commit_params& C = CP[k];
      C.n_instances = b->n; C.L = b->L; C.n_waves = b->n_waves; C.max_cycles = b->lim.max_cycles; C.wave_threads = (uint32_t)c->wave_width;
      C.cursors = b->d_cursors.p; C.dir = b->d_dir.p; C.scalars = b->d_scalars.p;
    }
    CP[0].queue = ZKW_QUEUE_LOG; CP[0].cap = b->cap_log; CP[0].per_instance_cap = per_log; CP[0].stream = b->d_log.p;
    CP[0].idx = b->d_ns_log_idx.p; CP[0].counts = b->d_ns_log_cnt.p;
    CP[1].queue = ZKW_QUEUE_DECOMMIT; CP[1].cap = b->cap_aux; CP[1].per_instance_cap = per_aux; CP[1].stream = b->d_auxs.p;
    CP[1].idx = b->d_ns_aux_idx.p; CP[1].counts = b->d_ns_aux_cnt.p; CP[1].aux_type_mask = 0xffffffffu;
    HIP_TRY(c, b->d_ns_bucket_params.alloc(2));
    HIP_TRY(c, hipMemcpy(b->d_ns_bucket_params.p, CP, sizeof CP, hipMemcpyHostToDevice));
    zkw_netstate_params N;
    std::memset(&N, 0, sizeof N);
    N.n_instances = b->n; N.L = b->L; N.n_waves = b->n_waves; N.max_cycles = b->lim.max_cycles; N.wave_threads = (uint32_t)c->wave_width;
    N.cap_log = b->cap_log; N.cap_aux = b->cap_aux; N.per_log = per_log; N.per_aux = per_aux; N.hist_cap = 2 * per_log; N.mark_cap = mark_cap;
    N.storage_aux_byte = c->isa.consts.storage_aux_byte; N.event_aux_byte = c->isa.consts.event_aux_byte; N.l1_aux_byte = c->isa.consts.l1_message_aux_byte;
    N.tails = b->d_tails.p; N.log_stream = b->d_log.p; N.aux_stream = b->d_auxs.p; N.scalars = b->d_scalars.p; N.scalars0 = b->d_scalars0.p;
    N.log_idx = b->d_ns_log_idx.p; N.log_cnt = b->d_ns_log_cnt.p; N.aux_idx = b->d_ns_aux_idx.p; N.aux_cnt = b->d_ns_aux_cnt.p;
    N.st_hist = b->d_ns_st_hist.p; N.ev_hist = b->d_ns_ev_hist.p; N.rb_st = b->d_ns_rb_st.p; N.rb_ev = b->d_ns_rb_ev.p; N.marks = b->d_ns_marks.p;
    N.out_counts = b->d_ns_counts.p;
    HIP_TRY(c, b->d_ns_params.alloc(1));
    HIP_TRY(c, hipMemcpy(b->d_ns_params.p, &N, sizeof N, hipMemcpyHostToDevice));
  }
  zkw_fused_table T;
  std::memset(&T, 0, sizeof T);
  T.n = 1; T.max_waves = b->n_waves; T.wave_threads = (uint32_t)c->wave_width;
  T.p[0] = b->d_ns_bucket_params.p;
  HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_BUCKET, st));
  T.p[0] = b->d_ns_bucket_params.p + 1;
  HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_BUCKET, st));
  T.p[0] = b->d_ns_params.p;
  HIP_TRY(c, zkw_launch_commit(&T, ZKW_COMMIT_STAGE_NETSTATE, st));
  b->run_stream = st;
  b->ns_done = true;
  b->ns_cached_wave = 0xffffffffu;
  return ZKW_OK;
}

// SimpleMemory::dump_page_content_as_u256_words (memory.rs:316-396)
int zkw_batch_get_page(zkw_batch* b, uint32_t instance, uint32_t page, uint32_t first_word, uint32_t n_words, zkw_u256* out) {
  if (!b || instance >= b->n || (n_words && !out)) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->uploaded) {
    c->last_error = "zkw_batch_get_page: batch not uploaded";
    return ZKW_ERR_INVALID;
  }
  int rc = zkw_batch_sync(b);
  if (rc != ZKW_OK) return rc;
  if (!n_words) return ZKW_OK;
  std::memset(out, 0, (size_t)n_words * sizeof(zkw_u256));
  const StagedInstance& si = b->staged[instance];
  const uint32_t F = b->lim.max_far_frames, L = b->L, w = instance / L, l = instance % L;
  auto from_vector = [&](const std::vector<zkw_u256>& v) {
    for (uint32_t k = 0; k < n_words; k++)
      if ((uint64_t)first_word + k < v.size()) out[k] = v[(size_t)first_word + k];
  };
  // 1. code pages (:321-332): page 0 (always present, all zero), the populated ones, the ones the run decommitted
  if (page == 0) return ZKW_OK;
  for (auto it = si.code_pages.rbegin(); it != si.code_pages.rend(); ++it)
    if (it->first == page) {
      from_vector(b->blobs[it->second]);
      return ZKW_OK;
    }
  {
    const zkw_dev_scalars& sc = b->h_scalars[instance];  // (zkw_batch_sync has refreshed them)
    const uint32_t pitch = b->kp.hist_pitch, np = (uint32_t)b->preimages.size();  // row p = (hash -> blob) pair p (zkw_dev_history)
    std::vector<zkw_dev_history> hist(sc.n_history ? np : 0);
    if (!hist.empty()) HIP_TRY(c, hipMemcpy(hist.data(), b->d_history.p + (size_t)instance * pitch, (size_t)np * sizeof(zkw_dev_history), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < hist.size(); k++)
      if (hist[k].valid && hist[k].page == page) {
        from_vector(b->blobs[b->preimages[k].second]);
        return ZKW_OK;
      }
  }
  // 2. pages with extended lifetime that are no arena pages: the bootloader's calldata (:229-231, 293-298)
  if (page == c->isa.consts.bootloader_calldata_page) {
    from_vector(si.bootloader_calldata);
    return ZKW_OK;
  }
  // 3. arena pages: returndata pages (extended lifetime), then the stack / heap / aux heap pages of live frames (:333-392)
  std::vector<zkw_dev_frame_meta> fm(F);
  HIP_TRY(c, hipMemcpy(fm.data(), b->d_frames.p + (size_t)instance * F, (size_t)F * sizeof(zkw_dev_frame_meta), hipMemcpyDeviceToHost));
  for (uint32_t slot = 0; slot < F; slot++) {
    const uint32_t st = fm[slot].stack_hwm, kind = page - fm[slot].base_page;
    if (st == 0xffffffffu || fm[slot].base_page == 0 || kind < 1 || kind > 3) continue;
    const bool live = (st & 0xc0000000u) == 0;                  // else kept / dead returndata: only that page has content
    if (!live && ((st >> 16) & 3u) != kind) return ZKW_OK;     // the frame's other pages went back to the pool: zeros
    const uint32_t hwm = kind == 1 ? st : (kind == 2 ? fm[slot].heap_hwm : fm[slot].aux_hwm);
    const uint32_t words = kind == 1 ? b->lim.stack_words : (kind == 2 ? b->lim.heap_words : b->lim.aux_heap_words);
    const uint4* arena = kind == 1 ? b->d_stack_vals.p : (kind == 2 ? b->d_heap.p : b->d_aux.p);
    const uint32_t end = std::min(std::min(hwm, words), first_word + n_words);
    if (first_word < end) {
      // a word is two 16-byte halves in two lane-minor planes of its row ([word][2][L]): rows of 16 bytes, pitch 16 L
      const uint4* src = arena + (((size_t)w * F + slot) * words + first_word) * 2 * L + l;
      HIP_TRY(c, hipMemcpy2DAsync(out, 16, src, (size_t)L * 16, 16, (size_t)(end - first_word) * 2, hipMemcpyDeviceToHost, b->run_stream));
      HIP_TRY(c, hipStreamSynchronize(b->run_stream));
    }
    return ZKW_OK;
  }
  return ZKW_OK;  // anything else: zeros (:395)
}

int zkw_batch_get_net_state(zkw_batch* b, uint32_t instance, zkw_net_state* out) {
  if (!b || !out || instance >= b->n) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  if (!b->ns_done) {
    int rc = zkw_batch_net_states(b, b->run_stream);
    if (rc != ZKW_OK) return rc;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(b->run_stream));
  std::memset(out, 0, sizeof *out);
  uint32_t cnt[4];
  HIP_TRY(c, hipMemcpy(cnt, b->d_ns_counts.p + (size_t)instance * 4, sizeof cnt, hipMemcpyDeviceToHost));
  if (cnt[3] & 2u) {
    c->last_error = "instance failed: no net state";
    return ZKW_ERR_INVALID;
  }
  if (cnt[3] & 1u) {
    c->last_error = "net state: per-instance index capacity exceeded (limits.max_log_queries / max_aux_events / max_callstack_depth)";
    return ZKW_ERR_LIMIT;
  }
  const uint32_t w = instance / b->L, per_log = b->lim.max_log_queries;
  if (b->ns_cached_wave != w) {  // the wave's log stream, once per wave
    uint32_t cur[4];
    HIP_TRY(c, hipMemcpy(cur, b->d_cursors.p + (size_t)w * 4, sizeof cur, hipMemcpyDeviceToHost));
    const uint32_t n_log = std::min(cur[1], b->cap_log);
    b->ns_wave_log.resize(n_log);
    if (n_log) HIP_TRY(c, hipMemcpy(b->ns_wave_log.data(), b->d_log.p + (size_t)w * b->cap_log * 8, (size_t)n_log * sizeof(zkw_log_query), hipMemcpyDeviceToHost));
    b->ns_cached_wave = w;
  }
  std::vector<uint32_t> sh(cnt[0]), eh(cnt[1]), ne(cnt[2]);
  if (cnt[0]) HIP_TRY(c, hipMemcpy(sh.data(), b->d_ns_st_hist.p + (size_t)instance * 2 * per_log, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost));
  if (cnt[1]) HIP_TRY(c, hipMemcpy(eh.data(), b->d_ns_ev_hist.p + (size_t)instance * 2 * per_log, (size_t)cnt[1] * 4, hipMemcpyDeviceToHost));
  if (cnt[2]) HIP_TRY(c, hipMemcpy(ne.data(), b->d_ns_rb_ev.p + (size_t)instance * per_log, (size_t)cnt[2] * 4, hipMemcpyDeviceToHost));
  auto materialise = [&](const std::vector<uint32_t>& idx, std::vector<zkw_log_query>& dst) {
    dst.resize(idx.size());
    for (size_t k = 0; k < idx.size(); k++) {
      zkw_log_query q = b->ns_wave_log[idx[k] & 0x7fffffffu];
      q.lane = 0; q.seq = 0; q.kind = 0;
      // the storage keeps a read as it arrived (written_value = 0: log.rs:175, far_call.rs:139); "written := read" is a
      // convention of access_storage for the witness tracer only (helpers.rs:143-146)
      if (!(q.bools & ZKW_LQ_RW)) std::memset(&q.written_value, 0, sizeof q.written_value);
      if (idx[k] & 0x80000000u) q.bools |= ZKW_LQ_ROLLBACK;
      dst[k] = q;
    }
  };
  materialise(sh, b->ns_st_hist);
  materialise(eh, b->ns_ev_hist);
  b->ns_events.clear();
  b->ns_l1.clear();
  for (uint32_t p : ne) {
    const zkw_log_query& q = b->ns_wave_log[p];
    zkw_event_message m;
    std::memset(&m, 0, sizeof m);
    m.shard_id = q.shard_id; m.is_first = (q.bools & ZKW_LQ_IS_SERVICE) ? 1 : 0; m.tx_number_in_block = q.tx_number_in_block;
    std::memcpy(m.address, q.address, 20);
    m.key = q.key; m.value = q.written_value;
    (q.aux_byte == c->isa.consts.event_aux_byte ? b->ns_events : b->ns_l1).push_back(m);
  }
  // final storage: the instance's table, entries that exist in the reference's `inner` map
  std::vector<zkw_dev_storage_entry> tab(b->lim.storage_slots);
  HIP_TRY(c, hipMemcpy(tab.data(), b->d_storage.p + (size_t)instance * b->lim.storage_slots, tab.size() * sizeof(zkw_dev_storage_entry), hipMemcpyDeviceToHost));
  b->ns_final.clear();
  for (const zkw_dev_storage_entry& e : tab) {
    if ((e.shard_state & 0x500u) != 0x500u) continue;
    zkw_storage_slot sl;
    std::memset(&sl, 0, sizeof sl);
    std::memcpy(sl.key.l, e.key, 32); std::memcpy(sl.value.l, e.value, 32);
    std::memcpy(sl.address, e.address, 20);  // same byte image as zkw_storage_slot / zkw_log_query
    sl.shard_id = (uint8_t)(e.shard_state & 0xffu);
    b->ns_final.push_back(sl);
  }
  std::sort(b->ns_final.begin(), b->ns_final.end(), [](const zkw_storage_slot& x, const zkw_storage_slot& y) {
    if (x.shard_id != y.shard_id) return x.shard_id < y.shard_id;
    int ca = std::memcmp(x.address, y.address, 20);
    if (ca) return ca < 0;
    for (int k = 3; k >= 0; k--)
      if (x.key.l[k] != y.key.l[k]) return x.key.l[k] < y.key.l[k];
    return false;
  });
  out->n_storage_history = (uint32_t)b->ns_st_hist.size(); out->n_event_history = (uint32_t)b->ns_ev_hist.size();
  out->n_events = (uint32_t)b->ns_events.size(); out->n_l1_messages = (uint32_t)b->ns_l1.size(); out->n_final_storage = (uint32_t)b->ns_final.size();
  out->storage_history = b->ns_st_hist.data(); out->event_history = b->ns_ev_hist.data(); out->events = b->ns_events.data();
  out->l1_messages = b->ns_l1.data(); out->final_storage = b->ns_final.data();
  return ZKW_OK;
}

int zkw_batch_get_commitments(zkw_batch* b, uint64_t* out) {
  if (!b || !out) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  int rc = zkw_batch_commit(b, 7u, b->run_stream);
  if (rc != ZKW_OK) return rc;
  HIP_TRY(c, hipStreamSynchronize(b->run_stream));
  HIP_TRY(c, hipMemcpy(out, b->d_commit.p, b->d_commit.bytes(), hipMemcpyDeviceToHost));
  return ZKW_OK;
}

int zkw_batch_step(zkw_batch* b, uint32_t max_cycles, uint32_t queue_mask, void* hip_stream) {
  if (!b) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  hipStream_t st = (hipStream_t)hip_stream;
  {
    const int grc = check_group(&b, 1);  // not uploaded / staged inputs changed since the upload: fail, never replay old inputs
    if (grc != ZKW_OK) return grc;
  }
  if (b->graph_exec && b->graph_cycles == max_cycles && b->graph_mask == queue_mask && b->graph_stream == st) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipGraphLaunch(b->graph_exec, st));
    b->cycles_run = max_cycles;
    b->ran = true;
    b->synced = false;
    b->wave_cache.clear();
    b->run_stream = st;
    b->pending_runs = 1;  // the captured run uses event pair 0
    b->dq_mode = b->graph_dq_mode;  // (the replayed reset + run, not whatever an earlier fused step left: a continued run / commit then agree with the device)
    b->ns_done = false;
    b->ns_cached_wave = 0xffffffffu;
    return ZKW_OK;
  }
  // eager pass first: validates arguments and performs every lazy allocation outside of stream capture
  int rc = zkw_batch_reset(b, hip_stream);
  if (rc == ZKW_OK) rc = zkw_batch_run(b, max_cycles, hip_stream);
  if (rc == ZKW_OK && queue_mask) rc = zkw_batch_commit(b, queue_mask, hip_stream);
  if (!b->graph_failed && c->opt_no_graph) b->graph_failed = true;  // diagnostics: eager steps only
  if (rc != ZKW_OK || b->graph_failed || !st) return rc;  // no capture on the legacy default stream
  // capture the same sequence for the following steps
  if (b->graph_exec) { (void)hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
  if (b->graph) { (void)hipGraphDestroy(b->graph); b->graph = nullptr; }
  HIP_TRY(c, hipStreamSynchronize(st));
  const uint32_t saved_pending = b->pending_runs;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) {
    b->graph_failed = true;
    return ZKW_OK;
  }
  b->pending_runs = 0;
  int crc = zkw_batch_reset(b, hip_stream);
  if (crc == ZKW_OK) crc = zkw_batch_run(b, max_cycles, hip_stream);
  if (crc == ZKW_OK && queue_mask) crc = zkw_batch_commit(b, queue_mask, hip_stream);
  hipGraph_t g = nullptr;
  const hipError_t e_end = hipStreamEndCapture(st, &g);
  b->pending_runs = saved_pending;
  if (crc != ZKW_OK || e_end != hipSuccess || !g) {
    if (g) (void)hipGraphDestroy(g);
    b->graph_failed = true;
    return ZKW_OK;
  }
  if (hipGraphInstantiate(&b->graph_exec, g, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(g);
    b->graph_exec = nullptr;
    b->graph_failed = true;
    return ZKW_OK;
  }
  b->graph = g;
  b->graph_dq_mode = b->dq_mode;
  b->graph_cycles = max_cycles;
  b->graph_mask = queue_mask;
  b->graph_stream = st;
  return ZKW_OK;
}

int zkw_batch_copy_commitments(zkw_batch* b, void* dst_device, void* hip_stream) {
  if (!b || !dst_device) return ZKW_ERR_INVALID;
  zkw_ctx* c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(dst_device, b->d_commit.p, b->d_commit.bytes(), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  return ZKW_OK;
}

int zkw_batch_commitments_device_ptr(zkw_batch* b, void** dptr, uint64_t* n_bytes) {
  if (!b || !dptr || !n_bytes) return ZKW_ERR_INVALID;
  *dptr = b->d_commit.p;
  *n_bytes = b->d_commit.bytes();
  return ZKW_OK;
}

// sizes of the ABI structs, for binding self-checks (tests/test_capi_layout.py)
uint32_t zkw_abi_sizeof(uint32_t which) {
  switch (which) {
    case 0: return sizeof(zkw_isa_table);
    case 1: return sizeof(zkw_callstack_entry);
    case 2: return sizeof(zkw_vm_local_state);
    case 3: return sizeof(zkw_block_properties);
    case 4: return sizeof(zkw_storage_slot);
    case 5: return sizeof(zkw_limits);
    case 6: return sizeof(zkw_cycle_record);
    case 7: return sizeof(zkw_mem_query);
    case 8: return sizeof(zkw_log_query);
    case 9: return sizeof(zkw_aux_event);
    case 10: return sizeof(zkw_instance_trace);
    case 11: return sizeof(zkw_run_stats);
    case 12: return sizeof(zkw_isa_consts);
    case 13: return sizeof(zkw_event_message);
    case 14: return sizeof(zkw_net_state);
    case 15: return sizeof(zkw_delivered);
    default: return 0;
  }
}

}  // extern "C"

// =================================================================================================
// Final exchange across the ranks of a job (include/zkw.h: zkw_comm_*, zkw_reduce_commitments).
// RCCL is loaded at run time (dlopen): libzkw.so does not link it, so the library loads on hosts without RCCL and a
// process that already carries an RCCL (e.g. the one inside a PyTorch wheel) keeps using that one.
// =================================================================================================
extern "C" hipError_t zkw_launch_pack_digests(const zkw_fused_table* T, uint64_t* dst, uint32_t n, uint32_t n_max, uint32_t mask, hipStream_t stream);

#ifndef ZKW_EMU_BUILD
#include <dlfcn.h>
namespace {
// the slice of the NCCL/RCCL C API this file uses (rccl.h: ncclUniqueId is 128 bytes, ncclComm_t an opaque pointer,
// ncclUint64 = 5, ncclSum = 0, ncclMax = 2)
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, zkw_comm_id, int) = nullptr;  // ncclUniqueId is passed by value
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
};
RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return &api;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (api.handle) break;
  }
  if (!api.handle) {
    api.error = "librccl.so not found (dlopen)";
    return &api;
  }
  api.GetUniqueId = (int (*)(void*))dlsym(api.handle, "ncclGetUniqueId");
  api.CommInitRank = (int (*)(void**, int, zkw_comm_id, int))dlsym(api.handle, "ncclCommInitRank");
  api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.handle, "ncclAllGather");
  api.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(api.handle, "ncclAllReduce");
  api.CommDestroy = (int (*)(void*))dlsym(api.handle, "ncclCommDestroy");
  api.GetErrorString = (const char* (*)(int))dlsym(api.handle, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.AllReduce || !api.CommDestroy) {
    api.error = "librccl.so lacks the ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclAllReduce symbols";
    api.handle = nullptr;
  }
  return &api;
}
}  // namespace
#endif

struct zkw_comm {
  zkw_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  void* nccl = nullptr;  // ncclComm_t (RCCL transport)
  zkw_allgather_fn ext_allgather = nullptr;
  zkw_allreduce_sum_u64_fn ext_allreduce = nullptr;
  void* ext_user = nullptr;
  // sizes of the ranks' shards: exchanged by the first zkw_reduce_commitments of the communicator and, after that, only
  // by zkw_comm_exchange_sizes (both collective)
  uint32_t sizes_for_n = 0xffffffffu, n_max = 0;
  std::vector<uint32_t> sizes;
  // packed digests of this rank, one buffer per stream the caller reduces on: the pack kernel and the all-gather of a
  // call are asynchronous, so calls on different streams must not share a send buffer (calls on one stream are ordered)
  std::map<hipStream_t, DevBuf<uint64_t>> d_send;
  DevBuf<uint64_t> d_small;          // staging of the small (host-synchronous) exchanges
  std::vector<uint64_t> h_send;      // external transport: host copy of the packed digests
};

static int comm_fail(zkw_comm* cm, const std::string& what) {
  cm->ctx->last_error = what;
  return ZKW_ERR_DEVICE;
}

// all-gather of `count` u64 per rank / all-reduce of `count` u64 on HOST buffers, over the communicator's transport
static int comm_allgather_host(zkw_comm* cm, const uint64_t* send, uint64_t* recv, uint32_t count, hipStream_t st) {
  if (cm->world == 1 && !cm->nccl) {  // (an RCCL communicator of one rank still goes through RCCL)
    std::memcpy(recv, send, (size_t)count * 8);
    return ZKW_OK;
  }
  if (cm->ext_allgather) return cm->ext_allgather(cm->ext_user, send, recv, (uint64_t)count * 8) == 0 ? ZKW_OK : comm_fail(cm, "external all-gather failed");
#ifndef ZKW_EMU_BUILD
  zkw_ctx* c = cm->ctx;
  RcclApi* api = rccl_api();
  if (cm->d_small.n < (size_t)count * (cm->world + 1)) {
    cm->d_small.release();
    HIP_TRY(c, cm->d_small.alloc((size_t)count * (cm->world + 1)));
  }
  HIP_TRY(c, hipMemcpyAsync(cm->d_small.p, send, (size_t)count * 8, hipMemcpyHostToDevice, st));
  const int e = api->AllGather(cm->d_small.p, cm->d_small.p + count, count, 5 /* ncclUint64 */, cm->nccl, st);
  if (e != 0) return comm_fail(cm, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(e) : "error"));
  HIP_TRY(c, hipMemcpyAsync(recv, cm->d_small.p + count, (size_t)count * cm->world * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  return ZKW_OK;
#else
  (void)st;
  return comm_fail(cm, "no transport");
#endif
}
static int comm_allreduce_host(zkw_comm* cm, uint64_t* inout, uint32_t count, bool max_op, hipStream_t st) {
  if (cm->world == 1 && !cm->nccl) return ZKW_OK;
  if (cm->ext_allreduce) {
    if (!max_op) return cm->ext_allreduce(cm->ext_user, inout, count) == 0 ? ZKW_OK : comm_fail(cm, "external all-reduce failed");
    // a maximum over the ranks through the all-gather (the external interface only offers a sum)
    std::vector<uint64_t> all((size_t)count * cm->world);
    int rc = comm_allgather_host(cm, inout, all.data(), count, st);
    if (rc != ZKW_OK) return rc;
    for (uint32_t i = 0; i < count; i++)
      for (int r = 0; r < cm->world; r++) inout[i] = std::max(inout[i], all[(size_t)r * count + i]);
    return ZKW_OK;
  }
#ifndef ZKW_EMU_BUILD
  zkw_ctx* c = cm->ctx;
  RcclApi* api = rccl_api();
  if (cm->d_small.n < count) {
    cm->d_small.release();
    HIP_TRY(c, cm->d_small.alloc(count));
  }
  HIP_TRY(c, hipMemcpyAsync(cm->d_small.p, inout, (size_t)count * 8, hipMemcpyHostToDevice, st));
  const int e = api->AllReduce(cm->d_small.p, cm->d_small.p, count, 5 /* ncclUint64 */, max_op ? 2 /* ncclMax */ : 0 /* ncclSum */, cm->nccl, st);
  if (e != 0) return comm_fail(cm, std::string("ncclAllReduce: ") + (api->GetErrorString ? api->GetErrorString(e) : "error"));
  HIP_TRY(c, hipMemcpyAsync(inout, cm->d_small.p, (size_t)count * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  return ZKW_OK;
#else
  (void)st;
  return comm_fail(cm, "no transport");
#endif
}

extern "C" {

// Can this process reach RCCL at all?  dlopen + symbol lookup only: no bootstrap root, no socket, no thread (ncclGetUniqueId
// starts all three — a probe through it left one listening root per rank for the life of the process).
int zkw_comm_probe(void) {
#ifndef ZKW_EMU_BUILD
  RcclApi* api = rccl_api();
  if (!api->handle) {
    g_create_error = api->error;
    return ZKW_ERR_DEVICE;
  }
  return ZKW_OK;
#else
  g_create_error = "RCCL is not part of the CPU emulation build";
  return ZKW_ERR_DEVICE;
#endif
}

int zkw_comm_get_unique_id(zkw_comm_id* out) {
  if (!out) return ZKW_ERR_INVALID;
#ifndef ZKW_EMU_BUILD
  RcclApi* api = rccl_api();
  if (!api->handle) {
    g_create_error = api->error;
    return ZKW_ERR_DEVICE;
  }
  static_assert(sizeof(zkw_comm_id) == 128, "ncclUniqueId");
  const int e = api->GetUniqueId(out);
  if (e != 0) {
    g_create_error = std::string("ncclGetUniqueId: ") + (api->GetErrorString ? api->GetErrorString(e) : "error");
    return ZKW_ERR_DEVICE;
  }
  return ZKW_OK;
#else
  g_create_error = "RCCL is not part of the CPU emulation build";
  return ZKW_ERR_DEVICE;
#endif
}

int zkw_comm_create_rccl(zkw_ctx* ctx, int rank, int world, const zkw_comm_id* id, zkw_comm** out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return ZKW_ERR_INVALID;
#ifndef ZKW_EMU_BUILD
  RcclApi* api = rccl_api();
  if (!api->handle) {
    ctx->last_error = api->error;
    return ZKW_ERR_DEVICE;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  auto* cm = new zkw_comm();
  cm->ctx = ctx; cm->rank = rank; cm->world = world;
  const int e = api->CommInitRank(&cm->nccl, world, *id, rank);
  if (e != 0) {
    ctx->last_error = std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(e) : "error");
    delete cm;
    return ZKW_ERR_DEVICE;
  }
  *out = cm;
  return ZKW_OK;
#else
  ctx->last_error = "RCCL is not part of the CPU emulation build";
  return ZKW_ERR_DEVICE;
#endif
}

int zkw_comm_create_external(zkw_ctx* ctx, int rank, int world, zkw_allgather_fn allgather, zkw_allreduce_sum_u64_fn allreduce_sum, void* user,
                             zkw_comm** out) {
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && (!allgather || !allreduce_sum))) return ZKW_ERR_INVALID;
  auto* cm = new zkw_comm();
  cm->ctx = ctx; cm->rank = rank; cm->world = world;
  cm->ext_allgather = allgather; cm->ext_allreduce = allreduce_sum; cm->ext_user = user;
  *out = cm;
  return ZKW_OK;
}

void zkw_comm_destroy(zkw_comm* cm) {
  if (!cm) return;
#ifndef ZKW_EMU_BUILD
  if (cm->nccl) (void)rccl_api()->CommDestroy(cm->nccl);
#endif
  for (auto& kv : cm->d_send) kv.second.release();
  cm->d_small.release();
  delete cm;
}

int zkw_comm_exchange_sizes(zkw_comm* cm, uint32_t n_instances, void* hip_stream) {
  if (!cm) return ZKW_ERR_INVALID;
  zkw_ctx* c = cm->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<uint64_t> all((size_t)cm->world);
  const uint64_t mine = n_instances;
  const int rc = comm_allgather_host(cm, &mine, all.data(), 1, (hipStream_t)hip_stream);
  if (rc != ZKW_OK) return rc;
  cm->sizes.assign(cm->world, 0);
  cm->n_max = 0;
  for (int r = 0; r < cm->world; r++) {
    cm->sizes[r] = (uint32_t)all[r];
    cm->n_max = std::max(cm->n_max, cm->sizes[r]);
  }
  cm->sizes_for_n = n_instances;
  return ZKW_OK;
}

int zkw_reduce_commitments(zkw_comm* cm, zkw_batch* const* batches, uint32_t n_batches, uint32_t queue_mask, void* gathered, uint32_t* n_max_out,
                           uint32_t* sizes_out, zkw_run_stats* total, void* hip_stream) {
  if (!cm || !batches || n_batches == 0) return ZKW_ERR_INVALID;
  int rc = check_group(batches, n_batches);
  if (rc != ZKW_OK) return rc;
  zkw_ctx* c = cm->ctx;
  if (batches[0]->ctx != c) {
    c->last_error = "zkw_reduce_commitments: the batches belong to another context than the communicator";
    return ZKW_ERR_INVALID;
  }
  const uint32_t n = batches[0]->n;
  for (uint32_t i = 1; i < n_batches; i++)
    if (batches[i]->n != n) {
      c->last_error = "zkw_reduce_commitments: the batches of one rank must have the same number of instances";
      return ZKW_ERR_INVALID;
    }
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(c, hipSetDevice(c->device));
  const bool external = !cm->nccl;
  // Shard sizes.  The FIRST reduce of a communicator exchanges them (every rank reaches its first reduce: a collective);
  // afterwards only zkw_comm_exchange_sizes does.  A rank whose n_instances changed on its own must not start an
  // exchange the other ranks do not take part in (their next collective would be a digest all-gather: a hang or garbage).
  if (cm->sizes_for_n == 0xffffffffu) {
    rc = zkw_comm_exchange_sizes(cm, n, hip_stream);
    if (rc != ZKW_OK) return rc;
  } else if (cm->sizes_for_n != n) {
    c->last_error = "zkw_reduce_commitments: n_instances differs from the shard size this communicator exchanged; call zkw_comm_exchange_sizes on EVERY rank first";
    return ZKW_ERR_INVALID;
  }
  if (n_max_out) *n_max_out = cm->n_max;
  if (sizes_out) std::memcpy(sizes_out, cm->sizes.data(), (size_t)cm->world * 4);
  queue_mask &= 7u;
  const uint32_t nq = (uint32_t)__builtin_popcount(queue_mask);
  if (gathered && nq) {
    const size_t per_rank = (size_t)n_batches * cm->n_max * nq * 4;  // u64
    DevBuf<uint64_t>& send = cm->d_send[st];
    if (send.n < per_rank) {
      if (send.p) HIP_TRY(c, hipStreamSynchronize(st));  // an earlier exchange of this stream may still read the old buffer
      send.release();
      HIP_TRY(c, send.alloc(per_rank));
    }
    zkw_fused_table T;
    std::memset(&T, 0, sizeof T);
    T.n = n_batches;
    T.wave_threads = (uint32_t)c->wave_width;
    for (uint32_t i = 0; i < n_batches; i++) T.p[i] = batches[i]->d_commit.p;
    HIP_TRY(c, zkw_launch_pack_digests(&T, send.p, n, cm->n_max, queue_mask, st));
    if (!external) {
#ifndef ZKW_EMU_BUILD
      RcclApi* api = rccl_api();
      const int e = api->AllGather(send.p, gathered, per_rank, 5 /* ncclUint64 */, cm->nccl, st);
      if (e != 0) return comm_fail(cm, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(e) : "error"));
#endif
    } else {  // host transport: the packed digests cross to the host, the callback exchanges them
      cm->h_send.resize(per_rank);
      HIP_TRY(c, hipMemcpyAsync(cm->h_send.data(), send.p, per_rank * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(c, hipStreamSynchronize(st));
      if (cm->world == 1) std::memcpy(gathered, cm->h_send.data(), per_rank * 8);
      else if (cm->ext_allgather(cm->ext_user, cm->h_send.data(), gathered, (uint64_t)per_rank * 8) != 0) return comm_fail(cm, "external all-gather failed");
    }
  }
  if (total) {
    uint64_t cnt[7] = {0, 0, 0, 0, 0, 0, 0};
    double kernel_ms = 0;
    for (uint32_t i = 0; i < n_batches; i++) {
      zkw_run_stats s;
      rc = zkw_batch_get_stats(batches[i], &s);  // waits for the batch's run
      if (rc != ZKW_OK) return rc;
      cnt[0] += s.cycles; cnt[1] += s.mem_queries; cnt[2] += s.log_queries; cnt[3] += s.aux_events;
      cnt[4] += s.instances_ended; cnt[5] += s.instances_failed; cnt[6] += s.reg_deltas;
      kernel_ms = std::max(kernel_ms, s.kernel_ms);
    }
    rc = comm_allreduce_host(cm, cnt, 7, false, st);
    if (rc != ZKW_OK) return rc;
    uint64_t ms_bits = (uint64_t)(kernel_ms * 1e6);  // ns, as an integer for the max-reduction
    rc = comm_allreduce_host(cm, &ms_bits, 1, true, st);
    if (rc != ZKW_OK) return rc;
    std::memset(total, 0, sizeof *total);
    total->cycles = cnt[0]; total->mem_queries = cnt[1]; total->log_queries = cnt[2]; total->aux_events = cnt[3];
    total->instances_ended = cnt[4]; total->instances_failed = cnt[5]; total->reg_deltas = cnt[6];
    total->kernel_ms = (double)ms_bits * 1e-6;
  }
  return ZKW_OK;
}

}  // extern "C"
