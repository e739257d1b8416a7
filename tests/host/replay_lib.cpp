// TEST INFRASTRUCTURE: drives the C++ host mirror (era-zk_evm_amd/host/zk_evm.hpp) over one
// instance trace with a tracer + event sink that write the canonical callback log of
// oracle/callback_log.hpp, so that tests can compare "GPU trace replayed through the mirror" with
// "what the oracle's restated cycle() called directly".
#include <vector>

#include "../../era-zk_evm_amd/host/zk_evm.hpp"
#include "../../oracle/callback_log.hpp"

using namespace zk_evm;

static void state_to_c(const VmLocalState& s, zkw_vm_local_state* o) {
  std::memset(o, 0, sizeof *o);
  std::memcpy(o->previous_code_word.l, s.previous_code_word.l, 32);
  uint16_t bm = 0;
  for (int i = 0; i < ZKW_REGISTERS_COUNT; i++) {
    std::memcpy(o->registers[i].l, s.registers[i].value.l, 32);
    if (s.registers[i].is_pointer) bm |= (uint16_t)(1u << i);
  }
  o->register_ptr_bitmap = bm;
  o->flags = (uint8_t)((s.flags.overflow_or_less_than_flag ? 1 : 0) | (s.flags.equality_flag ? 2 : 0) | (s.flags.greater_than_flag ? 4 : 0));
  o->pending_exception = s.pending_exception;
  o->previous_code_memory_page = s.previous_code_memory_page;
  o->timestamp = s.timestamp; o->monotonic_cycle_counter = s.monotonic_cycle_counter; o->spent_pubdata_counter = s.spent_pubdata_counter;
  o->memory_page_counter = s.memory_page_counter; o->absolute_execution_step = s.absolute_execution_step;
  o->current_ergs_per_pubdata_byte = s.current_ergs_per_pubdata_byte; o->tx_number_in_block = s.tx_number_in_block;
  o->previous_super_pc = s.previous_super_pc; o->callstack_depth = (uint32_t)s.callstack.depth();
  o->context_u128_register[0] = s.context_u128_register[0]; o->context_u128_register[1] = s.context_u128_register[1];
  o->current = s.callstack.current;
}
static cblog::MemQ cb_mem(const MemoryQuery& q) {
  cblog::MemQ m;
  m.timestamp = q.timestamp; m.page = q.location.page; m.index = q.location.index; m.type = q.location.memory_type; m.is_ptr = q.value_is_pointer;
  m.rw = q.rw_flag;
  std::memcpy(m.value, q.value.l, 32);
  return m;
}
static cblog::LogQ cb_log(const LogQuery& q) {
  cblog::LogQ l;
  l.timestamp = q.timestamp; l.tx = q.tx_number_in_block; l.aux = q.aux_byte; l.shard = q.shard_id; l.rw = q.rw_flag; l.rollback = q.rollback;
  l.is_service = q.is_service;
  std::memcpy(l.address, q.address.b, 20);
  std::memcpy(l.key, q.key.l, 32); std::memcpy(l.read, q.read_value.l, 32); std::memcpy(l.written, q.written_value.l, 32);
  return l;
}

struct LoggingTracer : VmWitnessTracer {
  cblog::Log* log;
  void st(uint32_t id, const VmLocalState& s) {
    zkw_vm_local_state c;
    state_to_c(s, &c);
    log->state(id, c, s.callstack.inner.data());
  }
  void start_new_execution_cycle(const VmLocalState& s) override { st(cblog::START_CYCLE, s); }
  void end_execution_cycle(const VmLocalState& s) override { st(cblog::END_CYCLE, s); }
  void add_memory_query(uint32_t cc, const MemoryQuery& q) override { log->mem(cc, cb_mem(q)); }
  void record_refund_for_query(uint32_t cc, const LogQuery& q, uint32_t) override { log->log(cblog::RECORD_REFUND, cc, cb_log(q)); }
  void add_log_query(uint32_t cc, const LogQuery& q) override { log->log(cblog::ADD_LOG_QUERY, cc, cb_log(q)); }
  void add_decommittment(uint32_t cc, const DecommittmentQuery& q, const std::vector<U256>& w) override {
    log->decommit(cc, q.hash.l, q.timestamp, q.memory_page, q.decommitted_length, q.is_fresh, (const uint64_t*)w.data(), w.size());
  }
  void add_precompile_call_result(uint32_t cc, const LogQuery& call, const std::vector<MemoryQuery>& in, const std::vector<MemoryQuery>& out,
                                  const PrecompileCyclesWitness& rw) override {
    std::vector<cblog::MemQ> a, b;
    for (auto& q : in) a.push_back(cb_mem(q));
    for (auto& q : out) b.push_back(cb_mem(q));
    std::vector<cblog::Log::Round> rounds;
    for (auto& r : rw.rounds) rounds.push_back(cblog::Log::Round{(uint8_t)r.has_new_request, (uint32_t)r.reads.size(), (uint32_t)r.writes.size()});
    log->precompile(cc, cb_log(call), a, b, (uint32_t)rw.kind, rounds);
  }
  void start_new_execution_context(uint32_t cc, const CallStackEntry& p, const CallStackEntry& n) override { log->frame_start(cc, p, n); }
  void finish_execution_context(uint32_t cc, bool panicked) override { log->simple(cblog::FINISH_CONTEXT, cc, panicked); }
};
struct LoggingSink : EventSink {
  cblog::Log* log;
  void add_partial_query(uint32_t cc, const LogQuery& q) override { log->log(cblog::EV_ADD_PARTIAL_QUERY, cc, cb_log(q)); }
  void start_frame(uint32_t ts) override { log->simple(cblog::EV_START_FRAME, ts, 0); }
  void finish_frame(bool panicked, uint32_t ts) override { log->simple(cblog::EV_FINISH_FRAME, panicked, ts); }
};

extern "C" int zkw_host_replay_callback_log(const zkw_vm_local_state* initial, const zkw_callstack_entry* inner, const zkw_instance_trace* trace,
                                            const zkw_u256* blob_words, const uint32_t* blob_first, const uint32_t* blob_len, uint32_t n_blobs,
                                            uint64_t* out, uint32_t cap, uint32_t* n_out, int* last_rc) {
  try {
    cblog::Log log;
    LoggingTracer wt;
    wt.log = &log;
    LoggingSink ev;
    ev.log = &log;
    BatchedVmState vm(*initial, inner, *trace, &wt, &ev);
    vm.code_of_blob = [&](uint32_t blob) {
      std::vector<U256> w;
      if (blob < n_blobs) {
        w.resize(blob_len[blob]);
        std::memcpy(w.data(), blob_words + blob_first[blob], (size_t)blob_len[blob] * 32);
      }
      return w;
    };
    int rc = 0;
    GenericNoopTracer debug_tracer;  // the caller's loop of cycle.rs:257-260: `vm.cycle(&mut tracer)?`
    while (!vm.execution_has_ended() && (rc = vm.cycle(debug_tracer)) == 0) {
    }
    *last_rc = rc;
    *n_out = (uint32_t)log.entries.size();
    for (uint32_t i = 0; i < log.entries.size() && i < cap; i++) out[i] = log.entries[i];
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}

// Replays a finished run into the host mirror of the reference's InMemoryEventSink and returns its flatten() result in
// the zkw_net_state conventions (history as zkw_log_query with lane / seq / kind = 0).
extern "C" int zkw_host_replay_event_sink(const zkw_vm_local_state* initial, const zkw_callstack_entry* inner, const zkw_instance_trace* trace,
                                          uint32_t initial_depth, uint8_t event_aux_byte, zkw_log_query* history, uint32_t cap_history, uint32_t* n_history,
                                          zkw_event_message* events, uint32_t cap_events, uint32_t* n_events, zkw_event_message* l1, uint32_t cap_l1,
                                          uint32_t* n_l1) {
  try {
    VmWitnessTracer wt;
    InMemoryEventSink ev;
    ev.event_aux_byte = event_aux_byte;
    for (uint32_t d = 0; d < initial_depth; d++) ev.start_frame(0);  // push_bootloader_context (helpers.rs:289-316)
    BatchedVmState vm(*initial, inner, *trace, &wt, &ev);
    int rc = 0;
    GenericNoopTracer debug_tracer;  // the caller's loop of cycle.rs:257-260: `vm.cycle(&mut tracer)?`
    while (!vm.execution_has_ended() && (rc = vm.cycle(debug_tracer)) == 0) {
    }
    std::vector<LogQuery> h;
    std::vector<EventMessage> e, m;
    ev.flatten(&h, &e, &m);
    *n_history = (uint32_t)h.size(); *n_events = (uint32_t)e.size(); *n_l1 = (uint32_t)m.size();
    for (uint32_t i = 0; i < h.size() && i < cap_history; i++) {
      zkw_log_query o;
      std::memset(&o, 0, sizeof o);
      const LogQuery& q = h[i];
      std::memcpy(o.key.l, q.key.l, 32); std::memcpy(o.read_value.l, q.read_value.l, 32); std::memcpy(o.written_value.l, q.written_value.l, 32);
      std::memcpy(o.address, q.address.b, 20);
      o.timestamp = q.timestamp; o.tx_number_in_block = q.tx_number_in_block; o.aux_byte = q.aux_byte; o.shard_id = q.shard_id;
      o.bools = (uint8_t)((q.rw_flag ? ZKW_LQ_RW : 0) | (q.rollback ? ZKW_LQ_ROLLBACK : 0) | (q.is_service ? ZKW_LQ_IS_SERVICE : 0));
      history[i] = o;
    }
    auto put = [](const EventMessage& x, zkw_event_message* o) {
      std::memset(o, 0, sizeof *o);
      o->shard_id = x.shard_id; o->is_first = x.is_first ? 1 : 0; o->tx_number_in_block = x.tx_number_in_block;
      std::memcpy(o->address, x.address.b, 20);
      std::memcpy(o->key.l, x.key.l, 32); std::memcpy(o->value.l, x.value.l, 32);
    };
    for (uint32_t i = 0; i < e.size() && i < cap_events; i++) put(e[i], &events[i]);
    for (uint32_t i = 0; i < m.size() && i < cap_l1; i++) put(m[i], &l1[i]);
    return rc == -1 || rc == 0 ? 0 : rc;  // -1 = the recorded cycles are exhausted (a run that was stopped while still running)
  } catch (const std::exception&) {
    return -1;
  }
}

// `vm.memory.dump_page_content(page, begin..end)` through the mirror (zk_evm::SimpleMemory over the C ABI's get_page of
// whichever library ran the batch): big-endian 32-byte words into `out`.
extern "C" int zkw_host_dump_page(void* get_page_fn, void* batch, uint32_t instance, uint32_t page, uint32_t begin, uint32_t end, uint8_t* out) {
  try {
    SimpleMemory m{(SimpleMemory::get_page_fn)get_page_fn, (zkw_batch*)batch, instance};
    const auto v = m.dump_page_content(page, begin, end);
    for (size_t i = 0; i < v.size(); i++) std::memcpy(out + 32 * i, v[i].data(), 32);
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}
