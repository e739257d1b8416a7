// ORACLE — TEST INFRASTRUCTURE ONLY (see vm.hpp header).
// C entry points of the CPU restatement.  They mirror include/zkw.h one-to-one with a
// `zkwo_` prefix so that tests drive the oracle and the HIP library through the same
// harness.  Instances run sequentially per thread ("one VmState per thread");
// zkwo_batch_set_threads picks the worker count for the cpu_baseline leg of bench.py.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include <pthread.h>
#include <sched.h>

#include "commit.hpp"
#include "vm.hpp"

using namespace zko;

struct zkwo_ctx {
  zkw_isa_table isa;
  bool has_isa = false;
  std::string last_error;
};

struct StagedInstance {
  zkw_vm_local_state state;
  std::vector<zkw_callstack_entry> inner;
  std::vector<std::pair<uint32_t, uint32_t>> code_pages;  // page -> blob
  std::vector<U256> heap;
  std::vector<U256> bootloader_calldata;
  std::vector<zkw_storage_slot> storage;
};

struct InstanceResult {
  uint32_t status = ZKW_STATUS_RUNNING;
  std::vector<zkw_log_query> ns_st_hist, ns_ev_hist;
  std::vector<zkw_event_message> ns_events, ns_l1;
  std::vector<zkw_storage_slot> ns_final;
  Recorder rec;
  zkw_vm_local_state final_state;
  std::string message;
  cblog::Log cb;
};

// Persistent, pinned worker threads for the cpu_baseline leg of bench.py: the timed region of a run then contains
// no thread creation, and every worker owns a contiguous block of instances ("one VmState per thread").
struct Pool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  uint64_t gen = 0;
  unsigned done = 0;
  bool stop = false;
  std::function<void(unsigned)> job;
  explicit Pool(const std::vector<int>& cpus, unsigned n) {
    for (unsigned t = 0; t < n; t++)
      th.emplace_back([this, t, cpus]() {
        if (!cpus.empty()) {
          cpu_set_t set;
          CPU_ZERO(&set);
          CPU_SET(cpus[t % cpus.size()], &set);
          (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
        }
        uint64_t seen = 0;
        for (;;) {
          std::unique_lock<std::mutex> lk(m);
          cv_go.wait(lk, [&] { return stop || gen != seen; });
          if (stop) return;
          seen = gen;
          lk.unlock();
          job(t);
          lk.lock();
          if (++done == th.size()) cv_done.notify_all();
        }
      });
  }
  void run(std::function<void(unsigned)> f) {
    std::unique_lock<std::mutex> lk(m);
    job = std::move(f);
    done = 0;
    gen++;
    cv_go.notify_all();
    cv_done.wait(lk, [&] { return done == th.size(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_go.notify_all();
    for (auto& x : th) x.join();
  }
};

struct zkwo_batch {
  std::unique_ptr<Pool> pool;
  ~zkwo_batch() { pool.reset(); }
  zkwo_ctx* ctx;
  uint32_t n;
  zkw_limits limits;
  std::vector<CodeBlob> blobs;
  std::unordered_map<U256, KnownCode, U256Hash> known_hashes;
  std::vector<StagedInstance> staged;
  zkw_block_properties props;
  std::vector<std::unique_ptr<Vm>> vms;
  std::vector<InstanceResult> results;
  unsigned threads = 1;
  bool callback_log = false;
  double last_ms = 0;
  bool ran = false;
};

static thread_local std::string g_create_error;

extern "C" {

int zkwo_ctx_create(int, zkwo_ctx** out) {
  *out = new zkwo_ctx();
  return ZKW_OK;
}
void zkwo_ctx_destroy(zkwo_ctx* c) { delete c; }
const char* zkwo_last_error(zkwo_ctx* c) { return c ? c->last_error.c_str() : g_create_error.c_str(); }
int zkwo_ctx_set_isa(zkwo_ctx* c, const zkw_isa_table* t) {
  c->isa = *t;
  c->has_isa = true;
  return ZKW_OK;
}

int zkwo_batch_create(zkwo_ctx* c, uint32_t n, const zkw_limits* limits, zkwo_batch** out) {
  if (!c->has_isa) {
    c->last_error = "set_isa first";
    return ZKW_ERR_INVALID;
  }
  auto* b = new zkwo_batch();
  b->ctx = c;
  b->n = n;
  b->limits = *limits;
  b->staged.resize(n);
  std::memset(&b->props, 0, sizeof b->props);
  b->blobs.push_back(std::make_shared<std::vector<U256>>());  // blob 0 = the all-zero page
  *out = b;
  return ZKW_OK;
}
void zkwo_batch_destroy(zkwo_batch* b) { delete b; }
int zkwo_batch_set_threads(zkwo_batch* b, uint32_t t) {
  b->threads = t ? t : 1;
  b->pool.reset();
  return ZKW_OK;
}
// `n_threads` persistent workers, worker t pinned to cpus[t % n_cpus] (n_cpus = 0: not pinned)
int zkwo_batch_set_pool(zkwo_batch* b, uint32_t n_threads, const int32_t* cpus, uint32_t n_cpus) {
  b->threads = n_threads ? n_threads : 1;
  b->pool.reset();
  if (b->threads > 1) b->pool.reset(new Pool(std::vector<int>(cpus, cpus + n_cpus), b->threads));
  return ZKW_OK;
}

int zkwo_batch_add_code_blob(zkwo_batch* b, const zkw_u256* words, uint32_t n_words, uint32_t* blob_id) {
  auto v = std::make_shared<std::vector<U256>>(n_words);
  if (n_words) std::memcpy(v->data(), words, (size_t)n_words * 32);
  b->blobs.push_back(v);
  *blob_id = (uint32_t)b->blobs.size() - 1;
  return ZKW_OK;
}
int zkwo_batch_add_decommit_preimage(zkwo_batch* b, const zkw_u256* hash, uint32_t blob_id) {
  if (blob_id >= b->blobs.size()) return ZKW_ERR_INVALID;
  U256 h;
  std::memcpy(h.l, hash->l, 32);
  if (b->known_hashes.count(h)) return ZKW_ERR_INVALID;  // decommitter.rs:25 assert
  const uint32_t index = (uint32_t)b->known_hashes.size();
  b->known_hashes[h] = KnownCode{blob_id, index, b->blobs[blob_id]};
  return ZKW_OK;
}
int zkwo_batch_set_code_page(zkwo_batch* b, uint32_t first, uint32_t count, uint32_t page, uint32_t blob_id) {
  if (first + count > b->n || blob_id >= b->blobs.size()) return ZKW_ERR_INVALID;
  for (uint32_t i = first; i < first + count; i++) b->staged[i].code_pages.emplace_back(page, blob_id);
  return ZKW_OK;
}
int zkwo_batch_set_state(zkwo_batch* b, uint32_t first, uint32_t count, const zkw_vm_local_state* states, const zkw_callstack_entry* inner,
                         uint32_t inner_depth) {
  if (first + count > b->n) return ZKW_ERR_INVALID;
  for (uint32_t i = 0; i < count; i++) {
    StagedInstance& s = b->staged[first + i];
    s.state = states[i];
    if (s.state.callstack_depth != inner_depth) return ZKW_ERR_INVALID;
    s.inner.assign(inner + (size_t)i * inner_depth, inner + (size_t)(i + 1) * inner_depth);
  }
  return ZKW_OK;
}
int zkwo_batch_set_heap(zkwo_batch* b, uint32_t instance, const zkw_u256* words, uint32_t n_words) {
  if (instance >= b->n) return ZKW_ERR_INVALID;
  b->staged[instance].heap.resize(n_words);
  if (n_words) std::memcpy(b->staged[instance].heap.data(), words, (size_t)n_words * 32);
  return ZKW_OK;
}
int zkwo_batch_set_bootloader_calldata(zkwo_batch* b, uint32_t instance, const zkw_u256* words, uint32_t n_words) {
  if (instance >= b->n) return ZKW_ERR_INVALID;
  b->staged[instance].bootloader_calldata.resize(n_words);
  if (n_words) std::memcpy(b->staged[instance].bootloader_calldata.data(), words, (size_t)n_words * 32);
  return ZKW_OK;
}
int zkwo_batch_set_storage(zkwo_batch* b, uint32_t instance, const zkw_storage_slot* slots, uint32_t n_slots) {
  if (instance >= b->n) return ZKW_ERR_INVALID;
  b->staged[instance].storage.assign(slots, slots + n_slots);
  return ZKW_OK;
}
int zkwo_batch_set_block_properties(zkwo_batch* b, const zkw_block_properties* p) {
  b->props = *p;
  return ZKW_OK;
}
int zkwo_batch_upload(zkwo_batch*) { return ZKW_OK; }

static void build_vm(zkwo_batch* b, uint32_t i) {
  const StagedInstance& s = b->staged[i];
  auto vm = std::make_unique<Vm>();
  vm->isa = &b->ctx->isa;
  std::memcpy(vm->block_properties.default_aa_code_hash.l, b->props.default_aa_code_hash.l, 32);
  vm->block_properties.zkporter_is_available = b->props.zkporter_is_available != 0;
  vm->decommittment_processor.known_hashes = &b->known_hashes;
  VmLocalState& L = vm->local_state;
  std::memcpy(L.previous_code_word.l, s.state.previous_code_word.l, 32);
  L.previous_code_memory_page = s.state.previous_code_memory_page;
  for (int r = 0; r < ZKW_REGISTERS_COUNT; r++) {
    std::memcpy(L.registers[r].value.l, s.state.registers[r].l, 32);
    L.registers[r].is_pointer = (s.state.register_ptr_bitmap >> r) & 1;
  }
  L.flags.overflow_or_less_than_flag = s.state.flags & 1;
  L.flags.equality_flag = s.state.flags & 2;
  L.flags.greater_than_flag = s.state.flags & 4;
  L.timestamp = s.state.timestamp;
  L.monotonic_cycle_counter = s.state.monotonic_cycle_counter;
  L.spent_pubdata_counter = s.state.spent_pubdata_counter;
  L.memory_page_counter = s.state.memory_page_counter;
  L.absolute_execution_step = s.state.absolute_execution_step;
  L.current_ergs_per_pubdata_byte = s.state.current_ergs_per_pubdata_byte;
  L.tx_number_in_block = s.state.tx_number_in_block;
  L.pending_exception = s.state.pending_exception != 0;
  L.previous_super_pc = s.state.previous_super_pc;
  L.context_u128_register[0] = s.state.context_u128_register[0];
  L.context_u128_register[1] = s.state.context_u128_register[1];
  entry_from_c(s.state.current, &L.callstack.current);
  L.callstack.inner.resize(s.inner.size());
  for (size_t d = 0; d < s.inner.size(); d++) entry_from_c(s.inner[d], &L.callstack.inner[d]);
  // what the host did before handing over: push_bootloader_context (helpers.rs:289-316) =
  // start_frame (storage/event-sink frames) + memory.start_global_frame for every far frame
  std::vector<const CallStackEntry*> frames;
  for (size_t d = 1; d < L.callstack.inner.size(); d++) frames.push_back(&L.callstack.inner[d]);
  if (!L.callstack.inner.empty()) frames.push_back(&L.callstack.current);
  for (const CallStackEntry* f : frames) {
    vm->storage.start_frame(0);
    vm->event_sink.start_frame(0);
    if (!f->is_local_frame) vm->memory.start_global_frame(0, f->base_memory_page, FatPointer::empty(), 0);
  }
  {  // zkw_batch_set_code_page: a later call for the same page replaces the earlier one
    std::unordered_map<uint32_t, uint32_t> last;
    for (auto& pb : s.code_pages) last[pb.first] = pb.second;
    for (auto& pb : last) vm->memory.populate_code(pb.first, *b->blobs[pb.second]);
  }
  if (!s.heap.empty()) vm->memory.populate_heap(s.heap);
  vm->memory.register_bootloader_calldata_page(b->ctx->isa.consts.bootloader_calldata_page);
  if (!s.bootloader_calldata.empty()) vm->memory.polulate_bootloaders_calldata(b->ctx->isa.consts.bootloader_calldata_page, s.bootloader_calldata);
  for (const zkw_storage_slot& sl : s.storage) {
    Address a;
    std::memcpy(a.b, sl.address, 20);
    U256 k, v;
    std::memcpy(k.l, sl.key.l, 32);
    std::memcpy(v.l, sl.value.l, 32);
    vm->storage.populate(sl.shard_id, a, k, v);
  }
  b->results[i] = InstanceResult();
  b->results[i].rec.init(L, b->pool ? b->limits.max_cycles : 0);  // pre-reserved only for the timed cpu_baseline runs
  b->results[i].rec.cb = b->callback_log ? &b->results[i].cb : nullptr;
  b->vms[i] = std::move(vm);
}

int zkwo_batch_reset(zkwo_batch* b, void*) {
  b->vms.clear();
  b->vms.resize(b->n);
  b->results.clear();
  b->results.resize(b->n);
  try {
    if (b->pool) {
      std::atomic<bool> failed{false};
      std::string msg;
      std::mutex mm;
      const unsigned T = (unsigned)b->pool->th.size();
      b->pool->run([&](unsigned t) {
        const uint32_t lo = (uint32_t)((uint64_t)b->n * t / T), hi = (uint32_t)((uint64_t)b->n * (t + 1) / T);
        try {
          for (uint32_t i = lo; i < hi; i++) build_vm(b, i);
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> lk(mm);
          failed = true;
          msg = e.what();
        }
      });
      if (failed) throw std::runtime_error(msg);
    } else {
      for (uint32_t i = 0; i < b->n; i++) build_vm(b, i);
    }
  } catch (const std::exception& e) {
    b->ctx->last_error = std::string("reset: ") + e.what();
    return ZKW_ERR_INVALID;
  }
  b->ran = false;
  return ZKW_OK;
}

static void run_instance(zkwo_batch* b, uint32_t i, uint32_t max_cycles) {
  Vm& vm = *b->vms[i];
  InstanceResult& r = b->results[i];
  std::swap(vm.witness_tracer, r.rec);
  uint32_t status = ZKW_STATUS_RUNNING;
  for (uint32_t k = 0; k < max_cycles; k++) {
    if (vm.local_state.execution_has_ended()) {
      status = ZKW_STATUS_ENDED;
      break;
    }
    try {
      vm.cycle();
    } catch (const RefErr& e) {
      status = ZKW_STATUS_UNKNOWN_CODE_HASH;
      r.message = e.what();
      vm.witness_tracer.rollback_cycle();
      break;
    } catch (const RefPanic& e) {
      status = ZKW_STATUS_REFERENCE_PANIC;
      r.message = e.what();
      vm.witness_tracer.rollback_cycle();
      break;
    }
  }
  if (status == ZKW_STATUS_RUNNING && vm.local_state.execution_has_ended()) status = ZKW_STATUS_ENDED;
  r.status = status;
  state_to_c(vm.local_state, &r.final_state);
  std::swap(vm.witness_tracer, r.rec);
}

int zkwo_batch_run(zkwo_batch* b, uint32_t max_cycles, void*) {
  if (b->vms.size() != b->n) return ZKW_ERR_INVALID;
  auto t0 = std::chrono::steady_clock::now();
  unsigned T = b->threads;
  if (T <= 1) {
    for (uint32_t i = 0; i < b->n; i++) run_instance(b, i, max_cycles);
  } else if (b->pool) {  // persistent pinned workers, one contiguous block of instances each
    const unsigned TP = (unsigned)b->pool->th.size();
    b->pool->run([&](unsigned t) {
      const uint32_t lo = (uint32_t)((uint64_t)b->n * t / TP), hi = (uint32_t)((uint64_t)b->n * (t + 1) / TP);
      for (uint32_t i = lo; i < hi; i++) run_instance(b, i, max_cycles);
    });
  } else {
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
      th.emplace_back([&]() {
        for (;;) {
          uint32_t i = next.fetch_add(1);
          if (i >= b->n) break;
          run_instance(b, i, max_cycles);
        }
      });
    for (auto& x : th) x.join();
  }
  b->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  b->ran = true;
  return ZKW_OK;
}
int zkwo_batch_sync(zkwo_batch*) { return ZKW_OK; }

int zkwo_batch_get_stats(zkwo_batch* b, zkw_run_stats* out) {
  if (!b->ran) return ZKW_ERR_NOT_RUN;
  std::memset(out, 0, sizeof *out);
  for (auto& r : b->results) {
    out->cycles += r.rec.records.size();
    out->mem_queries += r.rec.mem.size();
    out->log_queries += r.rec.log.size();
    out->aux_events += r.rec.aux.size();
    if (r.status == ZKW_STATUS_ENDED) out->instances_ended++;
    if (r.status >= ZKW_STATUS_UNKNOWN_CODE_HASH) out->instances_failed++;
  }
  out->kernel_ms = b->last_ms;
  return ZKW_OK;
}

// `vm.memory.dump_page_content_as_u256_words(page, first..first + n)` after the run (memory.rs:316-396)
int zkwo_batch_get_page(zkwo_batch* b, uint32_t instance, uint32_t page, uint32_t first_word, uint32_t n_words, zkw_u256* out) {
  if (!b->ran) return ZKW_ERR_NOT_RUN;
  if (instance >= b->n || b->vms.size() != b->n || !b->vms[instance]) return ZKW_ERR_INVALID;
  const std::vector<U256> v = b->vms[instance]->memory.dump_page_content_as_u256_words(page, first_word, n_words);
  for (uint32_t k = 0; k < n_words; k++) std::memcpy(out[k].l, v[k].l, 32);
  return ZKW_OK;
}

int zkwo_batch_get_instance_trace(zkwo_batch* b, uint32_t instance, zkw_instance_trace* out) {
  if (!b->ran) return ZKW_ERR_NOT_RUN;
  if (instance >= b->n) return ZKW_ERR_INVALID;
  InstanceResult& r = b->results[instance];
  std::memset(out, 0, sizeof *out);
  out->status = r.status;
  out->n_cycles = (uint32_t)r.rec.records.size();
  out->n_mem = (uint32_t)r.rec.mem.size();
  out->n_log = (uint32_t)r.rec.log.size();
  out->n_aux = (uint32_t)r.rec.aux.size();
  out->records = r.rec.records.data();
  out->mem = r.rec.mem.data();
  out->log = r.rec.log.data();
  out->aux = r.rec.aux.data();
  out->mem_off = r.rec.mem_off.data();
  out->log_off = r.rec.log_off.data();
  out->aux_off = r.rec.aux_off.data();
  out->final_state = r.final_state;
  return ZKW_OK;
}
// ---- final net states: get_final_net_states (testing/mod.rs:42-71) restated on the oracle's own sinks ----
static zkw_log_query net_log_to_c(const LogQuery& q) {
  zkw_log_query o;
  std::memset(&o, 0, sizeof o);
  std::memcpy(o.key.l, q.key.l, 32); std::memcpy(o.read_value.l, q.read_value.l, 32); std::memcpy(o.written_value.l, q.written_value.l, 32);
  std::memcpy(o.address, q.address.b, 20);
  o.timestamp = q.timestamp; o.tx_number_in_block = q.tx_number_in_block; o.aux_byte = q.aux_byte; o.shard_id = q.shard_id;
  o.bools = (uint8_t)((q.rw_flag ? ZKW_LQ_RW : 0) | (q.rollback ? ZKW_LQ_ROLLBACK : 0) | (q.is_service ? ZKW_LQ_IS_SERVICE : 0));
  return o;
}
static zkw_event_message net_event_to_c(const EventMessage& e) {
  zkw_event_message m;
  std::memset(&m, 0, sizeof m);
  m.shard_id = e.shard_id; m.is_first = e.is_first ? 1 : 0; m.tx_number_in_block = e.tx_number_in_block;
  std::memcpy(m.address, e.address.b, 20);
  std::memcpy(m.key.l, e.key.l, 32); std::memcpy(m.value.l, e.value.l, 32);
  return m;
}
int zkwo_batch_net_states(zkwo_batch* b, void*) { return b->ran ? ZKW_OK : ZKW_ERR_NOT_RUN; }
int zkwo_batch_get_net_state(zkwo_batch* b, uint32_t instance, zkw_net_state* out) {
  if (!b->ran) return ZKW_ERR_NOT_RUN;
  if (instance >= b->n) return ZKW_ERR_INVALID;
  InstanceResult& r = b->results[instance];
  if (r.status >= ZKW_STATUS_UNKNOWN_CODE_HASH) {
    b->ctx->last_error = "instance failed: no net state";
    return ZKW_ERR_INVALID;
  }
  Vm& vm = *b->vms[instance];
  std::memset(out, 0, sizeof *out);
  try {
    r.ns_st_hist.clear();
    for (const LogQuery& q : flatten_history(vm.storage.frames_stack)) r.ns_st_hist.push_back(net_log_to_c(q));
    std::vector<LogQuery> eh;
    std::vector<EventMessage> ev, l1;
    vm.event_sink.flatten(vm.isa->consts.event_aux_byte, &eh, &ev, &l1);
    r.ns_ev_hist.clear(); r.ns_events.clear(); r.ns_l1.clear();
    for (const LogQuery& q : eh) r.ns_ev_hist.push_back(net_log_to_c(q));
    for (const EventMessage& e : ev) r.ns_events.push_back(net_event_to_c(e));
    for (const EventMessage& e : l1) r.ns_l1.push_back(net_event_to_c(e));
  } catch (const RefPanic& e) {
    b->ctx->last_error = std::string("net state: ") + e.what();
    return ZKW_ERR_INVALID;
  }
  // final_storage_state = storage.inner.clone() (testing/mod.rs:58), in the canonical order of zkw.h
  r.ns_final.clear();
  for (const auto& kv : vm.storage.inner) {
    zkw_storage_slot sl;
    std::memset(&sl, 0, sizeof sl);
    std::memcpy(sl.key.l, kv.first.key.l, 32); std::memcpy(sl.value.l, kv.second.l, 32);
    std::memcpy(sl.address, kv.first.address.b, 20);
    sl.shard_id = kv.first.shard_id;
    r.ns_final.push_back(sl);
  }
  std::sort(r.ns_final.begin(), r.ns_final.end(), [](const zkw_storage_slot& x, const zkw_storage_slot& y) {
    if (x.shard_id != y.shard_id) return x.shard_id < y.shard_id;
    int ca = std::memcmp(x.address, y.address, 20);
    if (ca) return ca < 0;
    for (int k = 3; k >= 0; k--)
      if (x.key.l[k] != y.key.l[k]) return x.key.l[k] < y.key.l[k];
    return false;
  });
  out->n_storage_history = (uint32_t)r.ns_st_hist.size(); out->n_event_history = (uint32_t)r.ns_ev_hist.size();
  out->n_events = (uint32_t)r.ns_events.size(); out->n_l1_messages = (uint32_t)r.ns_l1.size(); out->n_final_storage = (uint32_t)r.ns_final.size();
  out->storage_history = r.ns_st_hist.data(); out->event_history = r.ns_ev_hist.data(); out->events = r.ns_events.data();
  out->l1_messages = r.ns_l1.data(); out->final_storage = r.ns_final.data();
  return ZKW_OK;
}

int zkwo_batch_enable_callback_log(zkwo_batch* b, int on) {
  b->callback_log = on != 0;
  return ZKW_OK;
}
int zkwo_batch_get_callback_log(zkwo_batch* b, uint32_t instance, const uint64_t** entries, uint32_t* n) {
  if (instance >= b->results.size()) return ZKW_ERR_INVALID;
  *entries = b->results[instance].cb.entries.data();
  *n = (uint32_t)b->results[instance].cb.entries.size();
  return ZKW_OK;
}
const char* zkwo_batch_instance_message(zkwo_batch* b, uint32_t instance) { return instance < b->results.size() ? b->results[instance].message.c_str() : ""; }

int zkwo_batch_get_commitments(zkwo_batch* b, uint64_t* out) {
  if (!b->ran) return ZKW_ERR_NOT_RUN;
  gl::Perm perm;
  std::vector<gl::Digest> blob_digests;
  for (auto& bl : b->blobs) blob_digests.push_back(gl::blob_digest(perm, (const zkw_u256*)bl->data(), bl->size()));
  for (uint32_t i = 0; i < b->n; i++) {
    const Recorder& r = b->results[i].rec;
    gl::Digest d[3] = {gl::mem_queue(perm, r.mem.data(), r.mem.size()), gl::log_queue(perm, r.log.data(), r.log.size()),
                       gl::decommit_queue(perm, r.aux.data(), r.aux.size(), blob_digests)};
    for (int q = 0; q < 3; q++) std::memcpy(out + ((size_t)i * 3 + q) * 4, d[q].v, 32);
  }
  return ZKW_OK;
}

// zkw_blake2s256 (include/zkw.h): the CPU restatement (hashes.hpp: blake2s256), message by message
int zkwo_blake2s256(zkwo_ctx*, const uint8_t* data, const uint64_t* offsets, uint32_t n_messages, uint8_t* digests) {
  for (uint32_t i = 0; i < n_messages; i++) {
    if (offsets[i + 1] < offsets[i]) return ZKW_ERR_INVALID;
    blake2s256(data + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), digests + 32 * (size_t)i);
  }
  return ZKW_OK;
}

// ---- unit-test hooks --------------------------------------------------------------------
// op: 0 add (out[0]=result, out[1].l[0]=of) 1 sub 2 mul (out[0]=low,out[1]=high) 3 div (q,r) 4 shl 5 shr (b.l[0]=n)
int zkwo_u256_op(int op, const zkw_u256* a, const zkw_u256* bb, zkw_u256* out) {
  U256 x, y;
  std::memcpy(x.l, a->l, 32);
  std::memcpy(y.l, bb->l, 32);
  std::memset(out, 0, 64);
  bool of = false;
  switch (op) {
    case 0: { U256 r = overflowing_add(x, y, of); std::memcpy(out[0].l, r.l, 32); out[1].l[0] = of; break; }
    case 1: { U256 r = overflowing_sub(x, y, of); std::memcpy(out[0].l, r.l, 32); out[1].l[0] = of; break; }
    case 2: { uint64_t t[8]; full_mul(x, y, t); std::memcpy(out[0].l, t, 64); break; }
    case 3: { if (y.is_zero()) return ZKW_ERR_INVALID; U256 q, r; div_mod(x, y, q, r); std::memcpy(out[0].l, q.l, 32); std::memcpy(out[1].l, r.l, 32); break; }
    case 4: { U256 r = shl(x, (uint32_t)y.l[0]); std::memcpy(out[0].l, r.l, 32); break; }
    case 5: { U256 r = shr(x, (uint32_t)y.l[0]); std::memcpy(out[0].l, r.l, 32); break; }
    default: return ZKW_ERR_INVALID;
  }
  return ZKW_OK;
}

// Runs one precompile call the way the reference's keccak256 test does
// (testing/tests/precompiles/keccak256.rs:74-142): a SimpleMemory with one extra heaps entry
// for `page` plus a Heap(1) indirection, input words pre-written to the heap, then
// execute_precompile; returns the word at `out_index` of the page and the query counts.
int zkwo_precompile_test(int which, uint32_t page, const zkw_u256* heap_words, uint32_t n_words, const zkw_u256* abi_key, uint32_t out_index,
                         zkw_u256* out_word, uint32_t* n_reads, uint32_t* n_writes) {
  try {
    SimpleMemory memory;
    memory.heaps.push_back(HeapPair{page, std::vector<U256>(1 << 10, U256::zero()), 0, {}});  // keccak256.rs:81-84
    memory.page_numbers_indirections[page] = Indirection{IND_HEAP, 1};                       // keccak256.rs:85-88
    for (uint32_t i = 0; i < n_words; i++) {  // pad_and_fill_memory keccak256.rs:39-69
      MemoryQuery q{0, MemoryLocation{ZKW_MEM_HEAP, page, i}, U256::zero(), false, true};
      std::memcpy(q.value.l, heap_words[i].l, 32);
      memory.execute_partial_query(1, q);
    }
    LogQuery lq;
    std::memset(&lq, 0, sizeof lq);
    lq.timestamp = 1;
    std::memcpy(lq.key.l, abi_key->l, 32);
    std::vector<MemoryQuery> reads, writes;
    std::vector<std::pair<uint32_t, uint32_t>> rounds;
    if (which == 0) keccak256_rounds_function(4, lq, memory, reads, writes, rounds);
    else if (which == 1) sha256_rounds_function(4, lq, memory, reads, writes, rounds);
    else ecrecover_function(4, lq, memory, (uint32_t)(which - 2), reads, writes, rounds);  // 2: (hash, r, s, v)   3: (hash, v, r, s)
    *n_reads = (uint32_t)reads.size();
    *n_writes = (uint32_t)writes.size();
    U256 w = SimpleMemory::get_or_zero(memory.heaps.back().heap, out_index);
    std::memcpy(out_word->l, w.l, 32);
    return ZKW_OK;
  } catch (const std::exception&) {
    return ZKW_ERR_INVALID;
  }
}

}  // extern "C"
