// Host-side mirror of the reference's caller-facing surface for the hot path, above the C ABI.
//
// The reference is Rust (`zk_evm` v1.4.1); this image has no Rust toolchain, so — as the task
// prescribes for a compiled-language reference — the host side is C++ with the reference's names
// and argument meaning:
//   VmLocalState / CallStackEntry / PrimitiveValue / Flags   src/vm_state/mod.rs:31-73, execution_stack.rs:6-30
//   MemoryQuery / LogQuery / DecommittmentQuery               field sets pinned by helpers.rs:26-32, log.rs:85-97, helpers.rs:171-177
//   VmWitnessTracer (10 callbacks)                             src/witness_trace/mod.rs:11-72
//   EventSink {add_partial_query, start_frame, finish_frame}   src/reference_impls/event_sink.rs:134-176
//   BatchedVmState::cycle(tracer) / execution_has_ended()      src/vm_state/cycle.rs:257-260, mod.rs:214-216
//   Tracer (debug hooks; accepted, const-gated off) / GenericNoopTracer   src/tracing.rs:40-72, utils.rs:51-92
//   SimpleMemory::dump_page_content_as_u256_words / dump_page_content     src/reference_impls/memory.rs:300-396
// `cycle()` is served from a finished GPU run: it replays one cycle's records of a
// zkw_instance_trace into the tracer and the event sink in the exact order the reference calls
// them and rebuilds the full VmLocalState the reference would pass to
// start_new_execution_cycle / end_execution_cycle.  Header-only, no GPU or HIP dependency.
#pragma once
#include <cstdint>
#include <array>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <map>
#include <vector>

#include "../../include/zkw.h"

namespace zk_evm {

struct U256 {
  uint64_t l[4];
};
struct Address {
  uint8_t b[20];  // little-endian integer bytes (zkw.h convention; H160 is the byte-reversed form)
};
struct PrimitiveValue {  // mod.rs:31-35
  U256 value;
  bool is_pointer;
};
struct Flags {  // flags.rs:4-8
  bool overflow_or_less_than_flag, equality_flag, greater_than_flag;
};
typedef zkw_callstack_entry CallStackEntry;  // execution_stack.rs:6-24, field for field
struct Callstack {                           // execution_stack.rs:27-30
  CallStackEntry current;
  std::vector<CallStackEntry> inner;
  size_t depth() const { return inner.size(); }
  bool is_empty() const { return inner.empty(); }
};
struct VmLocalState {  // mod.rs:54-73
  U256 previous_code_word;
  uint32_t previous_code_memory_page;
  PrimitiveValue registers[ZKW_REGISTERS_COUNT];
  Flags flags;
  uint32_t timestamp, monotonic_cycle_counter, spent_pubdata_counter, memory_page_counter, absolute_execution_step, current_ergs_per_pubdata_byte;
  uint16_t tx_number_in_block;
  bool pending_exception;
  uint16_t previous_super_pc;
  uint64_t context_u128_register[2];
  Callstack callstack;
  bool execution_has_ended() const { return callstack.is_empty(); }
};
struct MemoryLocation {
  uint8_t memory_type;  // ZKW_MEM_*
  uint32_t page, index;
};
struct MemoryQuery {
  uint32_t timestamp;
  MemoryLocation location;
  U256 value;
  bool value_is_pointer, rw_flag;
};
struct LogQuery {
  uint32_t timestamp;
  uint16_t tx_number_in_block;
  uint8_t aux_byte, shard_id;
  Address address;
  U256 key, read_value, written_value;
  bool rw_flag, rollback, is_service;
};
struct DecommittmentQuery {
  U256 hash;
  uint32_t timestamp, memory_page;
  uint16_t decommitted_length;
  bool is_fresh;
};

// PrecompileCyclesWitness (zk_evm_abstractions::precompiles, the 5th argument of add_precompile_call_result,
// witness_trace/mod.rs:43-50): an enum over the precompile kinds, each a Vec of per-round witnesses
// {new_request: Option<LogQuery>, reads: [..MemoryQuery..], writes: Option<[MemoryQuery; n]>}.  The crate is not on disk
// (recalled, unverified): one struct covers the three kinds here, `has_new_request` = Some/None, empty
// `writes` = None.  The rounds are rebuilt from the call's ordered read / write lists by the round schedule of the
// precompile (precompile_round_witness below) — the device emits every read exactly once, in round order.
struct PrecompileRoundWitness {
  bool has_new_request;
  LogQuery new_request;
  std::vector<MemoryQuery> reads, writes;
};
struct PrecompileCyclesWitness {
  enum Kind { Sha256 = 0, Keccak256 = 1, ECRecover = 2 } kind;
  std::vector<PrecompileRoundWitness> rounds;
};

// witness_trace/mod.rs:11-72 — same callbacks, same argument order
struct VmWitnessTracer {
  virtual ~VmWitnessTracer() {}
  virtual void start_new_execution_cycle(const VmLocalState&) {}
  virtual void end_execution_cycle(const VmLocalState&) {}
  virtual void add_memory_query(uint32_t /*monotonic_cycle_counter*/, const MemoryQuery&) {}
  virtual void record_refund_for_query(uint32_t, const LogQuery&, uint32_t /*pubdata refund; RefundType::None = 0*/) {}
  virtual void add_log_query(uint32_t, const LogQuery&) {}
  virtual void add_decommittment(uint32_t, const DecommittmentQuery&, const std::vector<U256>& /*mem_witness*/) {}
  virtual void add_precompile_call_result(uint32_t, const LogQuery& /*call_params*/, const std::vector<MemoryQuery>& /*mem_witness_in*/,
                                          const std::vector<MemoryQuery>& /*memory_witness_out*/, const PrecompileCyclesWitness& /*round_witness*/) {}
  virtual void add_revertable_precompile_call(uint32_t, const LogQuery&) {}  // never invoked by the reference
  virtual void start_new_execution_context(uint32_t, const CallStackEntry& /*previous*/, const CallStackEntry& /*new*/) {}
  virtual void finish_execution_context(uint32_t, bool /*panicked*/) {}
};

// event_sink.rs:134-176 (note the argument order of finish_frame)
struct EventSink {
  virtual ~EventSink() {}
  virtual void add_partial_query(uint32_t, const LogQuery&) {}
  virtual void start_frame(uint32_t /*timestamp*/) {}
  virtual void finish_frame(bool /*panicked*/, uint32_t /*timestamp*/) {}
};

// reference_impls/event_sink.rs:7-14
struct EventMessage {
  uint8_t shard_id;
  bool is_first;
  uint16_t tx_number_in_block;
  Address address;
  U256 key, value;
};

// The reference's own EventSink implementation (reference_impls/event_sink.rs:51-176), for callers that used
// `InMemoryEventSink` under the reference: driven by BatchedVmState's replay it ends in the same state, so `flatten()`
// (:66-131) returns the same (history, events, l1 messages) — the device computes the same thing for every instance
// at once (zkw_batch_get_net_state); tests/test_host_replay.py checks the two against each other.
struct InMemoryEventSink : EventSink {
  struct ApplicationData {  // :29-33
    std::vector<LogQuery> forward, rollbacks;
  };
  std::vector<ApplicationData> frames_stack;
  uint8_t event_aux_byte = 1;
  InMemoryEventSink() { frames_stack.emplace_back(); }  // :61: a single keeper frame
  void add_partial_query(uint32_t, const LogQuery& query) override {  // :140-151
    ApplicationData& f = frames_stack.back();
    f.forward.push_back(query);
    LogQuery rb = query;
    rb.rollback = true;
    f.rollbacks.push_back(rb);
  }
  void start_frame(uint32_t) override { frames_stack.emplace_back(); }  // :152-155
  void finish_frame(bool panicked, uint32_t) override {                  // :156-176
    ApplicationData cur = std::move(frames_stack.back());
    frames_stack.pop_back();
    ApplicationData& parent = frames_stack.back();
    parent.forward.insert(parent.forward.end(), cur.forward.begin(), cur.forward.end());
    if (panicked) parent.forward.insert(parent.forward.end(), cur.rollbacks.rbegin(), cur.rollbacks.rend());
    else parent.rollbacks.insert(parent.rollbacks.end(), cur.rollbacks.begin(), cur.rollbacks.end());
  }
  // :66-131; open frames (an instance that is still running) are netted as if they were kept
  void flatten(std::vector<LogQuery>* history, std::vector<EventMessage>* events, std::vector<EventMessage>* l1_messages) const {
    history->clear();
    for (const ApplicationData& f : frames_stack) history->insert(history->end(), f.forward.begin(), f.forward.end());
    std::map<uint32_t, LogQuery> tmp;
    for (const LogQuery& el : *history) {
      auto it = tmp.find(el.timestamp);
      if (it != tmp.end()) tmp.erase(it);  // :88 (a rollback of the entry with this timestamp)
      else tmp.emplace(el.timestamp, el);
    }
    events->clear();
    l1_messages->clear();
    for (const auto& kv : tmp) {
      const LogQuery& el = kv.second;
      EventMessage m{el.shard_id, el.is_service, el.tx_number_in_block, el.address, el.key, el.written_value};
      (el.aux_byte == event_aux_byte ? *events : *l1_messages).push_back(m);
    }
  }
};

// Splits the ordered reads / writes of one precompile call into its rounds.  sha256: `precompile_interpreted_data`
// rounds of two reads, the digest write in the last; ecrecover: one round (4 reads, 2 writes); keccak256: one round per
// 136-byte block (plus the padding-only round when the length is a multiple of 136), each reading — up to 6 words —
// what its buffer still lacks, the write in the last round.
inline PrecompileCyclesWitness precompile_round_witness(PrecompileCyclesWitness::Kind kind, const LogQuery& call, const std::vector<MemoryQuery>& in,
                                                        const std::vector<MemoryQuery>& out) {
  PrecompileCyclesWitness w;
  w.kind = kind;
  std::vector<std::pair<size_t, size_t>> plan;  // (reads, writes) per round
  const uint32_t in_off = (uint32_t)call.key.l[0], in_len = (uint32_t)(call.key.l[0] >> 32);
  if (kind == PrecompileCyclesWitness::Sha256) {
    const size_t rounds = (size_t)call.key.l[3];
    for (size_t r = 0; r < rounds; r++) plan.emplace_back(2, r + 1 == rounds ? 1 : 0);
  } else if (kind == PrecompileCyclesWitness::ECRecover) {
    plan.emplace_back(4, 2);
  } else {
    const size_t RATE = 136, PER_CYCLE = 6, BUF = PER_CYCLE * 32;
    size_t offset = in_off, left = in_len, rounds = (left + RATE - 1) / RATE, filled = 0;
    const bool extra = left % RATE == 0;
    if (extra) rounds++;
    for (size_t r = 0; r < rounds; r++) {
      const bool last = r + 1 == rounds, pad_only = extra && last;
      size_t n = 0;
      for (size_t i = 0; i < PER_CYCLE; i++) {
        const size_t at_most = 32 - offset % 32, meaningful = left >= at_most ? at_most : left;
        if (meaningful != 0 && !pad_only && filled + meaningful <= BUF) {
          offset += meaningful; left -= meaningful; filled += meaningful;
          n++;
        }
      }
      filled = filled < RATE ? 0 : filled - RATE;
      plan.emplace_back(n, last ? 1 : 0);
    }
  }
  size_t ri = 0, wi = 0;
  for (size_t r = 0; r < plan.size(); r++) {
    PrecompileRoundWitness rw;
    rw.has_new_request = r == 0;
    rw.new_request = call;
    for (size_t k = 0; k < plan[r].first && ri < in.size(); k++) rw.reads.push_back(in[ri++]);
    for (size_t k = 0; k < plan[r].second && wi < out.size(); k++) rw.writes.push_back(out[wi++]);
    w.rounds.push_back(std::move(rw));
  }
  return w;
}

inline U256 to_u256(const zkw_u256& v) {
  U256 r;
  std::memcpy(r.l, v.l, 32);
  return r;
}

// The debug tracer of `cycle<DT: Tracer>(&mut self, tracer: &mut DT)` (cycle.rs:257-260).  Its four hooks are compiled
// out unless the implementation's CALL_* constants say otherwise (tracing.rs:43-46); the replay has no decode internals
// to hand to them, so a tracer that enables one is refused at compile time.
struct GenericNoopTracer {  // utils.rs:51-92
  static constexpr bool CALL_BEFORE_DECODING = false, CALL_AFTER_DECODING = false, CALL_BEFORE_EXECUTION = false, CALL_AFTER_EXECUTION = false;
};

// `vm.memory` after the run, as far as callers read it (memory.rs:300-401).  `get_page` is the C-ABI entry of the
// library that ran the batch (zkw_batch_get_page of include/zkw.h).
struct SimpleMemory {
  typedef int (*get_page_fn)(zkw_batch*, uint32_t, uint32_t, uint32_t, uint32_t, zkw_u256*);
  get_page_fn get_page;
  zkw_batch* batch;
  uint32_t instance;
  // range = [begin, end) as in `std::ops::Range<u32>`
  std::vector<U256> dump_page_content_as_u256_words(uint32_t page_number, uint32_t begin, uint32_t end) const {  // :316-396
    std::vector<zkw_u256> raw(end > begin ? end - begin : 0);
    if (!raw.empty() && get_page(batch, instance, page_number, begin, (uint32_t)raw.size(), raw.data()) != ZKW_OK) throw std::runtime_error("zkw_batch_get_page failed");
    std::vector<U256> out(raw.size());
    for (size_t i = 0; i < raw.size(); i++) std::memcpy(out[i].l, raw[i].l, 32);
    return out;
  }
  std::vector<std::array<uint8_t, 32>> dump_page_content(uint32_t page_number, uint32_t begin, uint32_t end) const {  // :300-314: big-endian bytes
    const std::vector<U256> w = dump_page_content_as_u256_words(page_number, begin, end);
    std::vector<std::array<uint8_t, 32>> out(w.size());
    for (size_t i = 0; i < w.size(); i++)
      for (int b = 0; b < 32; b++) out[i][b] = (uint8_t)(w[i].l[3 - b / 8] >> (8 * (7 - b % 8)));
    return out;
  }
  std::vector<std::array<uint8_t, 32>> dump_full_page(uint32_t page_number) const { return dump_page_content(page_number, 0, 1u << 10); }  // :397-400
};

class BatchedVmState {
 public:
  VmLocalState local_state;
  VmWitnessTracer* witness_tracer;
  EventSink* event_sink;
  // code words for add_decommittment's mem_witness (blob id -> words); may be empty (then `B = false` behaviour: no payload)
  std::function<std::vector<U256>(uint32_t /*blob id*/)> code_of_blob;
  uint8_t event_aux_byte = 1, l1_message_aux_byte = 2, precompile_aux_byte = 3;  // system_params, log.rs:6-8
  // DefaultPrecompilesProcessor's dispatch on the low 16 address bits (zkw_isa_consts): calls to these are the ones
  // execute_precompile answers with Some(..), i.e. the ones add_precompile_call_result is invoked for (helpers.rs:210-221)
  uint32_t keccak_precompile_address = 0x8010, sha256_precompile_address = 0x02, ecrecover_precompile_address = 0x01;

  BatchedVmState(const zkw_vm_local_state& initial, const zkw_callstack_entry* inner, const zkw_instance_trace& trace, VmWitnessTracer* wt, EventSink* ev)
      : witness_tracer(wt), event_sink(ev), trace_(trace), k_(0) {
    VmLocalState& s = local_state;
    s.previous_code_word = to_u256(initial.previous_code_word);
    s.previous_code_memory_page = initial.previous_code_memory_page;
    for (int i = 0; i < ZKW_REGISTERS_COUNT; i++) s.registers[i] = PrimitiveValue{to_u256(initial.registers[i]), ((initial.register_ptr_bitmap >> i) & 1) != 0};
    s.flags = Flags{(initial.flags & 1) != 0, (initial.flags & 2) != 0, (initial.flags & 4) != 0};
    s.timestamp = initial.timestamp;
    s.monotonic_cycle_counter = initial.monotonic_cycle_counter;
    s.spent_pubdata_counter = initial.spent_pubdata_counter;
    s.memory_page_counter = initial.memory_page_counter;
    s.absolute_execution_step = initial.absolute_execution_step;
    s.current_ergs_per_pubdata_byte = initial.current_ergs_per_pubdata_byte;
    s.tx_number_in_block = initial.tx_number_in_block;
    s.pending_exception = initial.pending_exception != 0;
    s.previous_super_pc = initial.previous_super_pc;
    s.context_u128_register[0] = initial.context_u128_register[0];
    s.context_u128_register[1] = initial.context_u128_register[1];
    s.callstack.current = initial.current;
    s.callstack.inner.assign(inner, inner + initial.callstack_depth);
  }

  bool execution_has_ended() const { return local_state.execution_has_ended(); }  // mod.rs:214-216
  uint32_t cycles_available() const { return trace_.n_cycles - k_; }

  // VmState::cycle (cycle.rs:257-429). Returns 0 on success; ZKW_STATUS_* (>= 2) when the GPU run
  // stopped at this cycle with an error (the reference's Err / panic); -1 when the trace is exhausted.
  template <class DT>
  int cycle(DT& /*tracer*/) {
    static_assert(!(DT::CALL_BEFORE_DECODING || DT::CALL_AFTER_DECODING || DT::CALL_BEFORE_EXECUTION || DT::CALL_AFTER_EXECUTION),
                  "the replay serves finished cycles: the debug Tracer hooks (tracing.rs:40-72) are not available");
    return cycle();
  }
  int cycle() {
    if (k_ >= trace_.n_cycles) return trace_.status >= ZKW_STATUS_UNKNOWN_CODE_HASH ? (int)trace_.status : -1;
    VmLocalState& s = local_state;
    const uint32_t k = k_;
    const uint32_t cc = s.monotonic_cycle_counter;
    witness_tracer->start_new_execution_cycle(s);  // cycle.rs:34
    const CallStackEntry pre = s.callstack.current;
    const bool fetched = !s.pending_exception && (pre.code_page != s.previous_code_memory_page || (uint16_t)(pre.pc >> 2) != s.previous_super_pc);  // cycle.rs:58-60
    // merge the three streams of this cycle by their in-cycle sequence number
    uint32_t mi = trace_.mem_off[k], me = trace_.mem_off[k + 1], li = trace_.log_off[k], le = trace_.log_off[k + 1], ai = trace_.aux_off[k],
             ae = trace_.aux_off[k + 1];
    bool first_mem = true;
    bool in_precompile = false;
    LogQuery precompile_call{};
    std::vector<MemoryQuery> pin, pout;
    auto flush_precompile = [&]() {
      if (in_precompile) {  // every call to a known precompile, also one with no rounds (Some with empty vectors)
        const uint32_t low = (uint32_t)precompile_call.address.b[0] | ((uint32_t)precompile_call.address.b[1] << 8);
        int kind = low == keccak_precompile_address ? PrecompileCyclesWitness::Keccak256
                   : low == sha256_precompile_address ? PrecompileCyclesWitness::Sha256
                   : low == ecrecover_precompile_address ? PrecompileCyclesWitness::ECRecover : -1;
        if (kind >= 0)
          witness_tracer->add_precompile_call_result(cc, precompile_call, pin, pout,
                                                     precompile_round_witness((PrecompileCyclesWitness::Kind)kind, precompile_call, pin, pout));
      }
      in_precompile = false;
      pin.clear();
      pout.clear();
    };
    bool has_cold = false;
    zkw_aux_event cold{};
    while (mi < me || li < le || ai < ae) {
      // next record = smallest seq; ties (only possible at the saturated value 255): memory, then log, then aux
      int which = -1;
      uint32_t best = 0x7fffffff;
      if (mi < me && trace_.mem[mi].seq < best) { best = trace_.mem[mi].seq; which = 0; }
      if (li < le && trace_.log[li].seq < best) { best = trace_.log[li].seq; which = 1; }
      if (ai < ae && trace_.aux[ai].seq < best) { best = trace_.aux[ai].seq; which = 2; }
      if (which == 0) {
        const zkw_mem_query& r = trace_.mem[mi++];
        MemoryQuery q{r.timestamp, MemoryLocation{(uint8_t)(r.meta & ZKW_MQ_TYPE_MASK), r.page, r.index}, to_u256(r.value), (r.meta & ZKW_MQ_IS_PTR) != 0,
                      (r.meta & ZKW_MQ_RW) != 0};
        const uint32_t kind = r.meta >> ZKW_MQ_KIND_SHIFT;
        if (kind == 1) {
          pin.push_back(q);
        } else if (kind == 2) {
          pout.push_back(q);
        } else {
          flush_precompile();
          if (first_mem && fetched) s.previous_code_word = q.value;  // delayed_changes.new_previous_code_word (cycle.rs:83)
          witness_tracer->add_memory_query(cc, q);
        }
        first_mem = false;
      } else if (which == 1) {
        flush_precompile();
        const zkw_log_query& r = trace_.log[li++];
        LogQuery q;
        q.timestamp = r.timestamp; q.tx_number_in_block = r.tx_number_in_block; q.aux_byte = r.aux_byte; q.shard_id = r.shard_id;
        std::memcpy(q.address.b, r.address, 20);
        q.key = to_u256(r.key); q.read_value = to_u256(r.read_value); q.written_value = to_u256(r.written_value);
        q.rw_flag = (r.bools & ZKW_LQ_RW) != 0; q.rollback = (r.bools & ZKW_LQ_ROLLBACK) != 0; q.is_service = (r.bools & ZKW_LQ_IS_SERVICE) != 0;
        if (r.kind == ZKW_LQ_REFUND) {
          witness_tracer->record_refund_for_query(cc, q, 0);  // helpers.rs:128-132
        } else {
          if (q.aux_byte == event_aux_byte || q.aux_byte == l1_message_aux_byte) event_sink->add_partial_query(cc, q);  // helpers.rs:157-162
          witness_tracer->add_log_query(cc, q);
          if (q.aux_byte == precompile_aux_byte) {  // helpers.rs:207-222
            in_precompile = true;
            precompile_call = q;
          }
        }
      } else {
        flush_precompile();
        const zkw_aux_event& e = trace_.aux[ai++];
        if (e.type == ZKW_AUX_FRAME_START) {  // helpers.rs:225-246
          event_sink->start_frame(s.timestamp);
          witness_tracer->start_new_execution_context(cc, e.u.frame.previous, e.u.frame.next);
          s.callstack.inner.push_back(e.u.frame.previous);
          s.callstack.current = e.u.frame.next;
        } else if (e.type == ZKW_AUX_FRAME_FINISH) {  // helpers.rs:248-264
          event_sink->finish_frame(e.flag != 0, s.timestamp);
          witness_tracer->finish_execution_context(cc, e.flag != 0);
          if (s.callstack.inner.empty()) throw std::runtime_error("frame finish on an empty callstack");
          s.callstack.current = s.callstack.inner.back();
          s.callstack.inner.pop_back();
        } else if (e.type == ZKW_AUX_DECOMMIT) {  // helpers.rs:164-194
          DecommittmentQuery q{to_u256(e.u.hash), e.a, e.b, (uint16_t)(e.c & 0xffffu), e.flag != 0};
          if (code_of_blob) witness_tracer->add_decommittment(cc, q, q.is_fresh ? code_of_blob(e.c >> 16) : std::vector<U256>());
        } else if (e.type == ZKW_AUX_COLD_STATE) {
          has_cold = true;
          cold = e;
        }
      }
    }
    flush_precompile();
    // state after the cycle
    const zkw_cycle_record& rec = trace_.records[k];
    for (int i = 0; i < ZKW_REGISTERS_COUNT; i++) s.registers[i] = PrimitiveValue{to_u256(rec.registers[i]), ((rec.tail.register_ptr_bitmap >> i) & 1) != 0};
    s.flags = Flags{(rec.tail.flags & 1) != 0, (rec.tail.flags & 2) != 0, (rec.tail.flags & 4) != 0};
    s.pending_exception = (rec.tail.flags & 8) != 0;
    s.timestamp = rec.tail.timestamp;
    s.previous_super_pc = rec.tail.previous_super_pc;
    s.previous_code_memory_page = pre.code_page;  // cycle.rs:49
    s.monotonic_cycle_counter = cc + 1;           // cycle.rs:411
    if (s.callstack.depth() != rec.tail.callstack_depth) throw std::runtime_error("replay: callstack depth mismatch");
    CallStackEntry& cur = s.callstack.current;
    cur.pc = rec.tail.pc; cur.sp = rec.tail.sp; cur.ergs_remaining = rec.tail.ergs_remaining;
    cur.heap_bound = rec.tail.heap_bound; cur.aux_heap_bound = rec.tail.aux_heap_bound;
    if (has_cold) {
      s.spent_pubdata_counter = cold.a;
      s.current_ergs_per_pubdata_byte = cold.b;
      s.tx_number_in_block = (uint16_t)cold.c;
      s.context_u128_register[0] = cold.u.cold.context_u128_register[0];
      s.context_u128_register[1] = cold.u.cold.context_u128_register[1];
      s.memory_page_counter = cold.u.cold.memory_page_counter;
    }
    witness_tracer->end_execution_cycle(s);  // cycle.rs:413
    k_++;
    return 0;
  }

 private:
  zkw_instance_trace trace_;
  uint32_t k_;
};

}  // namespace zk_evm
