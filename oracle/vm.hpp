// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load it.
//
// CPU restatement of matter-labs/era-zk_evm (`zk_evm` v1.4.1) `VmState::cycle()` and of the
// in-repo oracle implementations it drives.  Single instance, sequential, mirrors the Rust
// control flow 1:1 (same order of oracle calls and witness emissions); every function cites
// the reference file:line it follows (paths relative to /root/reference/src).
//
// PARITY STATUS: the Rust crate cannot be built in this environment and
// its two load-bearing dependencies (zkevm_opcode_defs, zk_evm_abstractions @ branch
// v1.4.1) are not on disk.  Pinned: the keccak256 precompile against the reference's 8 live
// tests (tests/test_oracle_precompiles.py), U256 arithmetic against Python integers, the
// ALU/flag identities of SURVEY Appendix C/D.  Everything that depends on the absent
// crates' tables (variant numbering, prices, ABI bit layouts, sha256 precompile memory
// pattern) is "parity unpinned": the ISA table is an INPUT (zkw_isa_table), never hard-coded.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../include/zkw.h"
#include "callback_log.hpp"
#include "hashes.hpp"
#include "u256.hpp"

namespace zko {

// an assert!/unwrap/expect/panic! of the reference fired
struct RefPanic : std::runtime_error {
  explicit RefPanic(const std::string& m) : std::runtime_error(m) {}
};
// cycle() returned anyhow::Err (only decommitter.rs:54-56)
struct RefErr : std::runtime_error {
  explicit RefErr(const std::string& m) : std::runtime_error(m) {}
};
#define REF_ASSERT(c, msg) \
  do {                     \
    if (!(c)) throw RefPanic(msg); \
  } while (0)

// ---------------------------------------------------------------------------------------
// basic types
// ---------------------------------------------------------------------------------------

struct Address {  // 160-bit, little-endian bytes (zkw.h convention)
  uint8_t b[20];
  bool operator==(const Address& o) const { return std::memcmp(b, o.b, 20) == 0; }
};
inline Address address_zero() {
  Address a;
  std::memset(a.b, 0, 20);
  return a;
}
inline Address address_from_low_u32(uint32_t v) {
  Address a = address_zero();
  for (int i = 0; i < 4; i++) a.b[i] = (uint8_t)(v >> (8 * i));
  return a;
}
// utils.rs:36-41 address_to_u256
inline U256 address_to_u256(const Address& a) {
  U256 r = U256::zero();
  for (int i = 0; i < 20; i++) r.l[i / 8] |= (uint64_t)a.b[i] << (8 * (i % 8));
  return r;
}
// utils.rs:43-48 u256_to_address_unchecked (lowest 160 bits)
inline Address u256_to_address_unchecked(const U256& v) {
  Address a;
  for (int i = 0; i < 20; i++) a.b[i] = (uint8_t)(v.l[i / 8] >> (8 * (i % 8)));
  return a;
}
// execution_stack.rs:83-87 address < 2^16
inline bool address_is_kernel(const Address& a) {
  for (int i = 2; i < 20; i++)
    if (a.b[i]) return false;
  return true;
}

// mod.rs:31-35
struct PrimitiveValue {
  U256 value;
  bool is_pointer;
  static PrimitiveValue empty() { return PrimitiveValue{U256::zero(), false}; }
};

// zkevm_opcode_defs::FatPointer (absent crate; layout per SURVEY Appendix B: offset bits
// 0-31 — consistent with uma.rs:337-338 — page 32-63, start 64-95, length 96-127)
struct FatPointer {
  uint32_t offset, memory_page, start, length;
  static FatPointer empty() { return FatPointer{0, 0, 0, 0}; }
  static FatPointer from_u256(const U256& v) {
    return FatPointer{(uint32_t)v.l[0], (uint32_t)(v.l[0] >> 32), (uint32_t)v.l[1], (uint32_t)(v.l[1] >> 32)};
  }
  U256 to_u256() const { return U256{{(uint64_t)offset | ((uint64_t)memory_page << 32), (uint64_t)start | ((uint64_t)length << 32), 0, 0}}; }
  bool validate_in_bounds() const { return offset < length; }
  bool validate_as_slice() const { return offset <= length; }
};
// FatPointerValidationException bits
enum { FPV_OFFSET_IS_NOT_ZERO_WHEN_EXPECTED = 1, FPV_DEREF_BEYOND_HEAP_RANGE = 2 };
inline uint32_t fat_pointer_validate(const FatPointer& p, bool is_fresh) {
  uint32_t e = 0;
  if (is_fresh && p.offset != 0) e |= FPV_OFFSET_IS_NOT_ZERO_WHEN_EXPECTED;
  uint32_t s = p.start + p.length;
  if (s < p.start) e |= FPV_DEREF_BEYOND_HEAP_RANGE;
  return e;
}

// flags.rs:4-8
struct Flags {
  bool overflow_or_less_than_flag, equality_flag, greater_than_flag;
  void reset() { overflow_or_less_than_flag = equality_flag = greater_than_flag = false; }
};

// execution_stack.rs:6-24
struct CallStackEntry {
  Address this_address, msg_sender, code_address;
  uint32_t base_memory_page, code_page;
  uint16_t sp, pc, exception_handler_location;
  uint32_t ergs_remaining;
  uint8_t this_shard_id, caller_shard_id, code_shard_id;
  bool is_static, is_local_frame;
  uint64_t context_u128_value[2];
  uint32_t heap_bound, aux_heap_bound;
  bool is_kernel_mode() const { return address_is_kernel(this_address); }
  static uint32_t code_page_candidate_from_base(uint32_t b) { return b; }   // :67-69
  static uint32_t stack_page_from_base(uint32_t b) { return b + 1; }        // :71-73
  static uint32_t heap_page_from_base(uint32_t b) { return b + 2; }         // :75-77
  static uint32_t aux_heap_page_from_base(uint32_t b) { return b + 3; }     // :79-81
};

// execution_stack.rs:27-30, 90-139
struct Callstack {
  CallStackEntry current;
  std::vector<CallStackEntry> inner;
  void push_entry(const CallStackEntry& e) {
    inner.push_back(current);
    current = e;
  }
  CallStackEntry pop_entry() {
    REF_ASSERT(!inner.empty(), "callstack pop on empty");
    CallStackEntry old = current;
    current = inner.back();
    inner.pop_back();
    return old;
  }
  size_t depth() const { return inner.size(); }
  bool is_empty() const { return inner.empty(); }
};

// mod.rs:54-73
struct VmLocalState {
  U256 previous_code_word;
  uint32_t previous_code_memory_page;
  PrimitiveValue registers[ZKW_REGISTERS_COUNT];
  Flags flags;
  uint32_t timestamp, monotonic_cycle_counter, spent_pubdata_counter, memory_page_counter, absolute_execution_step,
      current_ergs_per_pubdata_byte;
  uint16_t tx_number_in_block;
  bool pending_exception;
  uint16_t previous_super_pc;
  uint64_t context_u128_register[2];
  Callstack callstack;
  bool execution_has_ended() const { return callstack.is_empty(); }  // mod.rs:96-98
};

struct MemoryLocation {
  uint8_t memory_type;  // ZKW_MEM_*
  uint32_t page, index;
};
struct MemoryQuery {  // helpers.rs:26-32
  uint32_t timestamp;
  MemoryLocation location;
  U256 value;
  bool value_is_pointer, rw_flag;
};
struct LogQuery {  // log.rs:85-97
  uint32_t timestamp;
  uint16_t tx_number_in_block;
  uint8_t aux_byte, shard_id;
  Address address;
  U256 key, read_value, written_value;
  bool rw_flag, rollback, is_service;
};
struct DecommittmentQuery {  // helpers.rs:171-177
  U256 hash;
  uint32_t timestamp, memory_page;
  uint16_t decommitted_length;
  bool is_fresh;
};

struct BlockProperties {  // block_properties/mod.rs:4-7
  U256 default_aa_code_hash;
  bool zkporter_is_available;
};

typedef std::shared_ptr<const std::vector<U256>> CodeBlob;

// ---------------------------------------------------------------------------------------
// Witness recorder = the VmWitnessTracer (witness_trace/mod.rs:11-72) that writes the
// canonical per-instance trace of include/zkw.h
// ---------------------------------------------------------------------------------------

void entry_to_c(const CallStackEntry& e, zkw_callstack_entry* o);
void entry_from_c(const zkw_callstack_entry& c, CallStackEntry* e);
void state_to_c(const VmLocalState& s, zkw_vm_local_state* o);

struct Recorder {
  std::vector<zkw_cycle_record> records;
  std::vector<zkw_mem_query> mem;
  std::vector<zkw_log_query> log;
  std::vector<zkw_aux_event> aux;
  std::vector<uint32_t> mem_off, log_off, aux_off;  // [n_cycles + 1]
  uint32_t seq = 0;
  cblog::Log* cb = nullptr;  // optional: canonical log of the outward calls (tests/test_host_replay.py)
  // cold fields as of the previous end_execution_cycle
  uint32_t cold_spent = 0, cold_ergs_pp = 0, cold_tx = 0, cold_mpc = 0;
  uint64_t cold_ctx[2] = {0, 0};

  // `cycles_hint` > 0: capacity for that many cycles is reserved up front, so that a timed run does not reallocate
  void init(const VmLocalState& s, uint32_t cycles_hint = 0) {
    records.clear(); mem.clear(); log.clear(); aux.clear();
    if (cycles_hint) {
      records.reserve(cycles_hint); mem.reserve(2 * (size_t)cycles_hint + 64);
      mem_off.reserve(cycles_hint + 1); log_off.reserve(cycles_hint + 1); aux_off.reserve(cycles_hint + 1);
    }
    mem_off.assign(1, 0); log_off.assign(1, 0); aux_off.assign(1, 0);
    capture_cold(s);
  }
  void capture_cold(const VmLocalState& s) {
    cold_spent = s.spent_pubdata_counter; cold_ergs_pp = s.current_ergs_per_pubdata_byte; cold_tx = s.tx_number_in_block;
    cold_mpc = s.memory_page_counter; cold_ctx[0] = s.context_u128_register[0]; cold_ctx[1] = s.context_u128_register[1];
  }
  uint8_t next_seq() {
    uint8_t s = seq > 255 ? 255 : (uint8_t)seq;
    seq++;
    return s;
  }
  // witness_trace/mod.rs:13 start_new_execution_cycle
  void log_state(uint32_t id, const VmLocalState& s) {
    zkw_vm_local_state c;
    state_to_c(s, &c);
    std::vector<zkw_callstack_entry> inner(s.callstack.inner.size());
    for (size_t d = 0; d < inner.size(); d++) entry_to_c(s.callstack.inner[d], &inner[d]);
    cb->state(id, c, inner.data());
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
  size_t cb_mark = 0;
  void start_new_execution_cycle(const VmLocalState& s) {
    seq = 0;
    if (cb) {
      cb_mark = cb->entries.size();
      log_state(cblog::START_CYCLE, s);
    }
  }
  // witness_trace/mod.rs:19 add_memory_query; kind 1/2 = payload of add_precompile_call_result (:43-50)
  void add_memory_query(const MemoryQuery& q, int kind = 0, uint32_t cc = 0) {
    if (cb && kind == 0) cb->mem(cc, cb_mem(q));
    zkw_mem_query o;
    std::memset(&o, 0, sizeof o);
    o.timestamp = q.timestamp; o.page = q.location.page; o.index = q.location.index;
    o.seq = next_seq();
    o.meta = (uint8_t)((q.location.memory_type & ZKW_MQ_TYPE_MASK) | (q.value_is_pointer ? ZKW_MQ_IS_PTR : 0) |
                       (q.rw_flag ? ZKW_MQ_RW : 0) | (kind << ZKW_MQ_KIND_SHIFT));
    std::memcpy(o.value.l, q.value.l, 32);
    mem.push_back(o);
  }
  // witness_trace/mod.rs:22-31 record_refund_for_query (kind REFUND) / :33 add_log_query
  void add_log(const LogQuery& q, int kind, uint32_t cc = 0) {
    if (cb) cb->log(kind == ZKW_LQ_REFUND ? cblog::RECORD_REFUND : cblog::ADD_LOG_QUERY, cc, cb_log(q));
    zkw_log_query o;
    std::memset(&o, 0, sizeof o);
    std::memcpy(o.key.l, q.key.l, 32); std::memcpy(o.read_value.l, q.read_value.l, 32); std::memcpy(o.written_value.l, q.written_value.l, 32);
    std::memcpy(o.address, q.address.b, 20);
    o.timestamp = q.timestamp; o.tx_number_in_block = q.tx_number_in_block; o.aux_byte = q.aux_byte; o.shard_id = q.shard_id;
    o.bools = (uint8_t)((q.rw_flag ? ZKW_LQ_RW : 0) | (q.rollback ? ZKW_LQ_ROLLBACK : 0) | (q.is_service ? ZKW_LQ_IS_SERVICE : 0));
    o.kind = (uint8_t)kind;
    o.seq = next_seq();
    log.push_back(o);
  }
  // witness_trace/mod.rs:61-68 start_new_execution_context
  void frame_start(const CallStackEntry& prev, const CallStackEntry& next, bool far, uint32_t cc = 0) {
    if (cb) {
      zkw_callstack_entry a, b;
      entry_to_c(prev, &a);
      entry_to_c(next, &b);
      cb->frame_start(cc, a, b);
    }
    zkw_aux_event e;
    std::memset(&e, 0, sizeof e);
    e.type = ZKW_AUX_FRAME_START; e.seq = next_seq(); e.flag = far ? 1 : 0;
    entry_to_c(prev, &e.u.frame.previous);
    entry_to_c(next, &e.u.frame.next);
    aux.push_back(e);
  }
  // witness_trace/mod.rs:70-71 finish_execution_context
  void frame_finish(bool panicked, uint32_t cc = 0) {
    if (cb) cb->simple(cblog::FINISH_CONTEXT, cc, panicked);
    zkw_aux_event e;
    std::memset(&e, 0, sizeof e);
    e.type = ZKW_AUX_FRAME_FINISH; e.seq = next_seq(); e.flag = panicked ? 1 : 0;
    aux.push_back(e);
  }
  // witness_trace/mod.rs:35-41 add_decommittment (recorded whether or not B, helpers.rs:185-191)
  void decommit(const DecommittmentQuery& q, uint32_t blob_id, uint32_t preimage_index, uint32_t cc = 0, const std::vector<U256>* words = nullptr) {
    if (cb) {  // SimpleDecommitter<true>: Some(values) when fresh, Some(vec![]) otherwise (decommitter.rs:43-47,81-96)
      const bool payload = q.is_fresh && words;
      cb->decommit(cc, q.hash.l, q.timestamp, q.memory_page, q.decommitted_length, q.is_fresh, payload ? (const uint64_t*)words->data() : nullptr,
                   payload ? words->size() : 0);
    }
    zkw_aux_event e;
    std::memset(&e, 0, sizeof e);
    e.type = ZKW_AUX_DECOMMIT; e.seq = next_seq(); e.flag = q.is_fresh ? 1 : 0;
    e.a = q.timestamp; e.b = q.memory_page; e.c = (uint32_t)q.decommitted_length | (blob_id << 16);
    std::memcpy(e.u.hash.l, q.hash.l, 32);
    e.u.decommit.preimage_index = preimage_index;
    aux.push_back(e);
  }
  // witness_trace/mod.rs:16 end_execution_cycle
  void end_execution_cycle(const VmLocalState& s) {
    if (cb) log_state(cblog::END_CYCLE, s);
    if (s.spent_pubdata_counter != cold_spent || s.current_ergs_per_pubdata_byte != cold_ergs_pp || s.tx_number_in_block != cold_tx ||
        s.memory_page_counter != cold_mpc || s.context_u128_register[0] != cold_ctx[0] || s.context_u128_register[1] != cold_ctx[1]) {
      zkw_aux_event e;
      std::memset(&e, 0, sizeof e);
      e.type = ZKW_AUX_COLD_STATE; e.seq = next_seq();
      e.a = s.spent_pubdata_counter; e.b = s.current_ergs_per_pubdata_byte; e.c = s.tx_number_in_block;
      e.u.cold.context_u128_register[0] = s.context_u128_register[0]; e.u.cold.context_u128_register[1] = s.context_u128_register[1];
      e.u.cold.memory_page_counter = s.memory_page_counter;
      aux.push_back(e);
      capture_cold(s);
    }
    zkw_cycle_record r;
    std::memset(&r, 0, sizeof r);
    uint16_t bitmap = 0;
    for (int i = 0; i < ZKW_REGISTERS_COUNT; i++) {
      std::memcpy(r.registers[i].l, s.registers[i].value.l, 32);
      if (s.registers[i].is_pointer) bitmap |= (uint16_t)(1u << i);
    }
    const CallStackEntry& c = s.callstack.current;
    r.tail.register_ptr_bitmap = bitmap;
    r.tail.flags = (uint8_t)((s.flags.overflow_or_less_than_flag ? 1 : 0) | (s.flags.equality_flag ? 2 : 0) | (s.flags.greater_than_flag ? 4 : 0) |
                             (s.pending_exception ? 8 : 0));
    r.tail.pc = c.pc; r.tail.sp = c.sp; r.tail.ergs_remaining = c.ergs_remaining; r.tail.timestamp = s.timestamp;
    r.tail.heap_bound = c.heap_bound; r.tail.aux_heap_bound = c.aux_heap_bound;
    r.tail.callstack_depth = (uint16_t)s.callstack.depth(); r.tail.previous_super_pc = s.previous_super_pc;
    uint32_t nm = (uint32_t)mem.size() - mem_off.back(), nl = (uint32_t)log.size() - log_off.back(), na = (uint32_t)aux.size() - aux_off.back();
    r.tail.event_counts = (nm > 255 ? 255 : nm) | ((nl > 255 ? 255 : nl) << 8) | ((na > 255 ? 255 : na) << 16);
    records.push_back(r);
    mem_off.push_back((uint32_t)mem.size()); log_off.push_back((uint32_t)log.size()); aux_off.push_back((uint32_t)aux.size());
  }
  // a cycle that ended in RefPanic/RefErr leaves no trace
  void rollback_cycle() {
    if (cb) cb->entries.resize(cb_mark);
    mem.resize(mem_off.back()); log.resize(log_off.back()); aux.resize(aux_off.back());
  }
};

// ---------------------------------------------------------------------------------------
// SimpleMemory — reference_impls/memory.rs:149-759
// (pools :79-147 and pre-allocation :214-241 are allocation strategy, not semantics: pages are
//  created empty and read as zero exactly like pooled, zero-filled pages do)
// ---------------------------------------------------------------------------------------

enum IndKind { IND_HEAP, IND_AUX_HEAP, IND_RETURNDATA_EXTENDED_LIFETIME, IND_EMPTY };
struct Indirection {  // memory.rs:69-75
  IndKind kind;
  size_t index;
};
struct HeapPair {
  uint32_t heap_page;
  std::vector<U256> heap;
  uint32_t aux_page;
  std::vector<U256> aux;
};

struct SimpleMemory {
  std::vector<std::pair<uint32_t, std::vector<PrimitiveValue>>> stack_pages;  // :151
  std::vector<HeapPair> heaps;                                                // :152
  std::unordered_map<uint32_t, std::vector<U256>> code_pages;                 // :153 (zero-extended to 2^16 words on read)
  std::unordered_map<uint32_t, std::vector<U256>> pages_with_extended_lifetime;  // :164
  std::unordered_map<uint32_t, Indirection> page_numbers_indirections;           // :165
  std::vector<std::unordered_set<uint32_t>> indirections_to_cleanup_on_return;   // :166

  static const uint32_t MAX_STACK_PAGE_SIZE_IN_WORDS = 1u << 16;
  static const uint32_t MAX_CODE_PAGE_SIZE_IN_WORDS = 1u << 16;

  SimpleMemory() {  // new_without_preallocations :243-266
    code_pages[0] = std::vector<U256>();
    page_numbers_indirections[0] = Indirection{IND_EMPTY, 0};
    indirections_to_cleanup_on_return.emplace_back();
    heaps.push_back(HeapPair{0, {}, 0, {}});
  }
  // :271-284
  void populate_code(uint32_t page, const std::vector<U256>& values) {
    REF_ASSERT(code_pages.find(page) == code_pages.end() || page == 0, "populate_code: page exists");
    REF_ASSERT(values.size() <= MAX_CODE_PAGE_SIZE_IN_WORDS, "populate_code: too long");
    code_pages[page] = values;
  }
  // :287-291
  void populate_heap(const std::vector<U256>& values) { heaps.back().heap = values; }
  // BOOTLOADER_CALLDATA_PAGE lives in `pages_with_extended_lifetime` from the start (:229-231, 257-259; the constant is an
  // ISA-table input here) and polulate_bootloaders_calldata replaces its content (:293-298).  No indirection is
  // registered for it (:233, 261 insert page 0 only): the VM cannot read it, `dump_page_content` can.
  void register_bootloader_calldata_page(uint32_t page) { pages_with_extended_lifetime[page]; }
  void polulate_bootloaders_calldata(uint32_t page, const std::vector<U256>& values) {
    REF_ASSERT(pages_with_extended_lifetime.find(page) != pages_with_extended_lifetime.end(), "bootloader calldata page missing");  // :296 unwrap
    pages_with_extended_lifetime[page] = values;
  }
  // dump_page_content_as_u256_words (:316-396): code pages, pages with extended lifetime, live stack pages (values only),
  // live heaps — in that order; anything else reads as zero
  std::vector<U256> dump_page_content_as_u256_words(uint32_t page_number, uint32_t first, uint32_t n) const {
    std::vector<U256> result(n, U256::zero());
    auto take = [&](const std::vector<U256>& content) {
      for (uint32_t k = 0; k < n; k++) result[k] = get_or_zero(content, (size_t)first + k);
      return result;
    };
    auto cp = code_pages.find(page_number);
    if (cp != code_pages.end()) return take(cp->second);  // :321-332
    auto ext = pages_with_extended_lifetime.find(page_number);
    if (ext != pages_with_extended_lifetime.end()) return take(ext->second);  // :334-345
    for (auto it = stack_pages.rbegin(); it != stack_pages.rend(); ++it) {  // :347-362
      if (it->first != page_number) continue;
      for (uint32_t k = 0; k < n; k++)
        if ((size_t)first + k < it->second.size()) result[k] = it->second[(size_t)first + k].value;
      return result;
    }
    for (auto it = heaps.rbegin(); it != heaps.rend(); ++it) {  // :364-392
      if (it->heap_page == page_number) return take(it->heap);
      if (it->aux_page == page_number) return take(it->aux);
    }
    return result;  // :395
  }

  static void resize_to_fit(std::vector<U256>& el, size_t idx) {  // :194-200
    if (el.size() >= idx + 1) return;
    el.resize(idx + 1, U256::zero());
  }
  static U256 get_or_zero(const std::vector<U256>& v, size_t idx) { return idx < v.size() ? v[idx] : U256::zero(); }

  // :404-528
  MemoryQuery execute_partial_query(uint32_t, MemoryQuery query) {
    uint32_t page_number = query.location.page;
    switch (query.location.memory_type) {
      case ZKW_MEM_STACK: {
        REF_ASSERT(!stack_pages.empty(), "no stack page");
        auto& top = stack_pages.back();
        REF_ASSERT(top.first == page_number, "stack page mismatch");                          // :415,428
        REF_ASSERT(query.location.index < MAX_STACK_PAGE_SIZE_IN_WORDS, "out of bounds for stack page");  // :420-424
        size_t idx = query.location.index;
        if (query.rw_flag) {
          if (top.second.size() <= idx) top.second.resize(idx + 1, PrimitiveValue::empty());
          top.second[idx] = PrimitiveValue{query.value, query.value_is_pointer};
        } else {
          PrimitiveValue p = idx < top.second.size() ? top.second[idx] : PrimitiveValue::empty();
          query.value = p.value;
          query.value_is_pointer = p.is_pointer;
        }
        break;
      }
      case ZKW_MEM_HEAP:
      case ZKW_MEM_AUX_HEAP: {
        REF_ASSERT(query.value_is_pointer == false, "heap value is pointer");  // :440
        HeapPair& cur = heaps.back();
        // :447,451,463,467 are debug_assert_eq on the page number: compiled out in release
        std::vector<U256>& content = query.location.memory_type == ZKW_MEM_HEAP ? cur.heap : cur.aux;
        resize_to_fit(content, query.location.index);  // reads grow too, :464,468
        if (query.rw_flag)
          content[query.location.index] = query.value;
        else
          query.value = content[query.location.index];
        break;
      }
      case ZKW_MEM_FAT_PTR: {
        REF_ASSERT(query.rw_flag == false, "fat pointer write");           // :476
        REF_ASSERT(query.value_is_pointer == false, "fat pointer is ptr");  // :477
        auto it = page_numbers_indirections.find(page_number);
        REF_ASSERT(it != page_numbers_indirections.end(), "fat pointer only points to reachable memory");  // :478-481
        const Indirection& ind = it->second;
        switch (ind.kind) {
          case IND_HEAP: {
            const HeapPair& f = heaps[ind.index];
            REF_ASSERT(f.heap_page == query.location.page, "indirection heap page mismatch");  // :489
            query.value = get_or_zero(f.heap, query.location.index);
            break;
          }
          case IND_AUX_HEAP: {
            const HeapPair& f = heaps[ind.index];
            REF_ASSERT(f.aux_page == query.location.page, "indirection aux page mismatch");  // :499
            query.value = get_or_zero(f.aux, query.location.index);
            break;
          }
          case IND_RETURNDATA_EXTENDED_LIFETIME: {
            auto p = pages_with_extended_lifetime.find(page_number);
            REF_ASSERT(p != pages_with_extended_lifetime.end(), "indirection target must exist");  // :508-511
            query.value = get_or_zero(p->second, query.location.index);
            break;
          }
          case IND_EMPTY:
            query.value = U256::zero();
            break;
        }
        break;
      }
      default:
        throw RefPanic("code should be through specialized query");  // :522-524
    }
    return query;
  }
  // :530-554
  MemoryQuery specialized_code_query(uint32_t, MemoryQuery query) {
    REF_ASSERT(query.location.memory_type == ZKW_MEM_CODE, "not code");
    uint32_t page = query.location.page;
    size_t idx = query.location.index;
    if (query.rw_flag) {
      auto& content = code_pages[page];  // inserts a zero page if absent (:541-544)
      if (content.size() <= idx) content.resize(idx + 1, U256::zero());
      content[idx] = query.value;
    } else {
      auto it = code_pages.find(page);
      REF_ASSERT(it != code_pages.end(), "code page absent");
      query.value = get_or_zero(it->second, idx);
    }
    return query;
  }
  // :556-569
  MemoryQuery read_code_query(uint32_t, MemoryQuery query) const {
    REF_ASSERT(query.location.memory_type == ZKW_MEM_CODE, "not code");
    REF_ASSERT(!query.rw_flag, "code write through read_code_query");
    auto it = code_pages.find(query.location.page);
    REF_ASSERT(it != code_pages.end(), "code page absent");  // :565 unwrap
    query.value = get_or_zero(it->second, query.location.index);
    return query;
  }
  // :573-657
  void start_global_frame(uint32_t /*current_base_page*/, uint32_t new_base_page, FatPointer calldata_fat_pointer, uint32_t /*ts*/) {
    stack_pages.emplace_back(CallStackEntry::stack_page_from_base(new_base_page), std::vector<PrimitiveValue>());
    uint32_t heap_page = CallStackEntry::heap_page_from_base(new_base_page);
    uint32_t aux_heap_page = CallStackEntry::aux_heap_page_from_base(new_base_page);
    uint32_t current_heap_page = heaps.back().heap_page;
    uint32_t current_aux_heap_page = heaps.back().aux_page;
    size_t idx_to_use_for_calldata_ptrs = heaps.size() - 1;
    heaps.push_back(HeapPair{heap_page, {}, aux_heap_page, {}});
    indirections_to_cleanup_on_return.emplace_back();
    if (calldata_fat_pointer.memory_page == 0) {
    } else if (calldata_fat_pointer.memory_page == current_heap_page) {
      page_numbers_indirections[current_heap_page] = Indirection{IND_HEAP, idx_to_use_for_calldata_ptrs};
      indirections_to_cleanup_on_return.back().insert(current_heap_page);
    } else if (calldata_fat_pointer.memory_page == current_aux_heap_page) {
      page_numbers_indirections[current_aux_heap_page] = Indirection{IND_AUX_HEAP, idx_to_use_for_calldata_ptrs};
      indirections_to_cleanup_on_return.back().insert(current_aux_heap_page);
    } else {
      auto it = page_numbers_indirections.find(calldata_fat_pointer.memory_page);
      REF_ASSERT(it != page_numbers_indirections.end(), "fat pointer must only point to reachable memory");  // :642-645
      REF_ASSERT(it->second.kind == IND_HEAP || it->second.kind == IND_AUX_HEAP, "calldata forwarding without heap indirection");  // :646-655
    }
  }
  // :660-758
  void finish_global_frame(uint32_t base_page, FatPointer returndata_fat_pointer, uint32_t /*ts*/) {
    uint32_t stack_page = CallStackEntry::stack_page_from_base(base_page);
    REF_ASSERT(!stack_pages.empty(), "no stack page to pop");
    REF_ASSERT(stack_pages.back().first == stack_page, "stack page mismatch on finish");  // :673
    stack_pages.pop_back();
    uint32_t returndata_page = returndata_fat_pointer.memory_page;
    uint32_t heap_page = CallStackEntry::heap_page_from_base(base_page);
    uint32_t aux_heap_page = CallStackEntry::aux_heap_page_from_base(base_page);
    REF_ASSERT(!heaps.empty(), "no heaps");
    HeapPair cur = std::move(heaps.back());
    heaps.pop_back();
    REF_ASSERT(heap_page == cur.heap_page, "heap page mismatch");     // :690
    REF_ASSERT(aux_heap_page == cur.aux_page, "aux page mismatch");   // :691
    REF_ASSERT(!indirections_to_cleanup_on_return.empty(), "indirections must exist");  // :693-696
    std::unordered_set<uint32_t> current_cleanup = std::move(indirections_to_cleanup_on_return.back());
    indirections_to_cleanup_on_return.pop_back();
    REF_ASSERT(!indirections_to_cleanup_on_return.empty(), "previous page indirections must exist");  // :697-700
    std::unordered_set<uint32_t>& previous_cleanup = indirections_to_cleanup_on_return.back();
    if (returndata_page == cur.heap_page) {
      REF_ASSERT(pages_with_extended_lifetime.find(cur.heap_page) == pages_with_extended_lifetime.end(), "existing extended page");  // :707
      pages_with_extended_lifetime[cur.heap_page] = std::move(cur.heap);
      page_numbers_indirections[cur.heap_page] = Indirection{IND_RETURNDATA_EXTENDED_LIFETIME, 0};
      previous_cleanup.insert(cur.heap_page);
    } else if (returndata_page == cur.aux_page) {
      REF_ASSERT(pages_with_extended_lifetime.find(cur.aux_page) == pages_with_extended_lifetime.end(), "existing extended page");  // :719
      pages_with_extended_lifetime[cur.aux_page] = std::move(cur.aux);
      page_numbers_indirections[cur.aux_page] = Indirection{IND_RETURNDATA_EXTENDED_LIFETIME, 0};
      previous_cleanup.insert(cur.aux_page);
    } else {
      if (returndata_page != 0) {
        REF_ASSERT(page_numbers_indirections.find(returndata_page) != page_numbers_indirections.end(), "expected that indirections contain page");  // :734-742
        current_cleanup.erase(returndata_page);
        previous_cleanup.insert(returndata_page);
      }
    }
    for (uint32_t el : current_cleanup) {
      size_t n = page_numbers_indirections.erase(el);
      REF_ASSERT(n == 1, "double free in indirection");  // :755-756
    }
  }
};

// ---------------------------------------------------------------------------------------
// InMemoryStorage — testing/storage.rs:8-186;  InMemoryEventSink — reference_impls/event_sink.rs
// ---------------------------------------------------------------------------------------

struct StorageKey {
  uint8_t shard_id;
  Address address;
  U256 key;
  bool operator==(const StorageKey& o) const { return shard_id == o.shard_id && address == o.address && key == o.key; }
};
struct StorageKeyHash {
  size_t operator()(const StorageKey& k) const {
    uint64_t h = 0x9e3779b97f4a7c15ULL * (k.shard_id + 1);
    for (int i = 0; i < 4; i++) h = (h ^ k.key.l[i]) * 0xff51afd7ed558ccdULL, h ^= h >> 33;
    uint64_t a;
    std::memcpy(&a, k.address.b, 8);
    h = (h ^ a) * 0xc4ceb9fe1a85ec53ULL;
    return (size_t)(h ^ (h >> 29));
  }
};
struct ApplicationData {  // event_sink.rs:29-33
  std::vector<LogQuery> forward, rollbacks;
};

struct InMemoryStorage {
  std::unordered_map<StorageKey, U256, StorageKeyHash> inner;             // :9
  std::unordered_set<StorageKey, StorageKeyHash> cold_warm_markers;       // :10
  std::vector<ApplicationData> frames_stack;                              // :11
  InMemoryStorage() { frames_stack.emplace_back(); }
  void populate(uint8_t shard, const Address& a, const U256& key, const U256& value) { inner[StorageKey{shard, a, key}] = value; }  // :25-31
  // :80-86: RefundType::None => pubdata_refund() == 0
  uint32_t estimate_refunds_for_write(uint32_t, const LogQuery&) { return 0; }
  // :88-139
  LogQuery execute_partial_query(uint32_t, LogQuery query) {
    REF_ASSERT(!frames_stack.empty(), "frame must be started");
    ApplicationData& frame_data = frames_stack.back();
    REF_ASSERT(!query.rollback, "query.rollback");
    StorageKey k{query.shard_id, query.address, query.key};
    auto it = inner.find(k);
    U256 current_value = it == inner.end() ? U256::zero() : it->second;
    if (query.rw_flag) {
      inner[k] = query.written_value;
      cold_warm_markers.insert(k);
      query.read_value = current_value;
      frame_data.forward.push_back(query);
      query.rollback = true;
      frame_data.rollbacks.push_back(query);
      query.rollback = false;
    } else {
      if (it == inner.end()) {
        // `entry(address).or_default()` creates the address map only; the slot itself stays absent
      }
      cold_warm_markers.insert(k);
      query.read_value = current_value;
      frame_data.forward.push_back(query);
    }
    return query;
  }
  void start_frame(uint32_t) { frames_stack.emplace_back(); }  // :140-143
  // :144-186
  void finish_frame(uint32_t, bool panicked) {
    REF_ASSERT(!frames_stack.empty(), "frame must be started before finishing");
    ApplicationData current_frame = std::move(frames_stack.back());
    frames_stack.pop_back();
    REF_ASSERT(!frames_stack.empty(), "parent_frame_must_exist");
    ApplicationData& parent = frames_stack.back();
    if (panicked) {
      for (auto q = current_frame.rollbacks.rbegin(); q != current_frame.rollbacks.rend(); ++q) {
        auto it = inner.find(StorageKey{q->shard_id, q->address, q->key});
        REF_ASSERT(it != inner.end(), "must always exist on rollback");
        REF_ASSERT(it->second == q->written_value, "rollback value mismatch");  // :171
        it->second = q->read_value;
      }
      parent.forward.insert(parent.forward.end(), current_frame.forward.begin(), current_frame.forward.end());
      parent.forward.insert(parent.forward.end(), current_frame.rollbacks.rbegin(), current_frame.rollbacks.rend());
    } else {
      parent.forward.insert(parent.forward.end(), current_frame.forward.begin(), current_frame.forward.end());
      parent.rollbacks.insert(parent.rollbacks.end(), current_frame.rollbacks.begin(), current_frame.rollbacks.end());
    }
  }
};

// InMemoryStorage::flatten_and_net_history (testing/storage.rs:34-76), first component: the keeper frame's `forward`.
// The reference asserts frames_stack.len() == 1; for an instance that is still running the open frames are
// concatenated bottom-up, which is what the keeper's `forward` would become if every open frame were kept
// (storage.rs:181-185) — identical to the reference whenever it does not panic.
inline std::vector<LogQuery> flatten_history(const std::vector<ApplicationData>& frames_stack) {
  std::vector<LogQuery> history;
  for (const ApplicationData& f : frames_stack) history.insert(history.end(), f.forward.begin(), f.forward.end());
  return history;
}

struct EventMessage {  // event_sink.rs:7-14
  uint8_t shard_id;
  bool is_first;
  uint16_t tx_number_in_block;
  Address address;
  U256 key, value;
};

struct InMemoryEventSink {  // event_sink.rs:51-56, 134-176
  std::vector<ApplicationData> frames_stack;
  InMemoryEventSink() { frames_stack.emplace_back(); }
  void add_partial_query(uint32_t, LogQuery query, uint8_t event_aux, uint8_t l1_aux) {
    REF_ASSERT(query.rw_flag, "event rw_flag");
    REF_ASSERT(query.aux_byte == event_aux || query.aux_byte == l1_aux, "event aux byte");
    REF_ASSERT(!query.rollback, "event rollback");
    REF_ASSERT(!frames_stack.empty(), "frame must be started");
    ApplicationData& f = frames_stack.back();
    f.forward.push_back(query);
    query.rollback = true;
    f.rollbacks.push_back(query);
  }
  // flatten (event_sink.rs:66-131): net the forward history by timestamp, then split by aux byte
  void flatten(uint8_t event_aux, std::vector<LogQuery>* history, std::vector<EventMessage>* events, std::vector<EventMessage>* l1_messages) const {
    *history = flatten_history(frames_stack);
    std::map<uint32_t, LogQuery> tmp;  // :81 HashMap<u32, LogQuery>; the keys are sorted afterwards (:99-100)
    for (const LogQuery& el : *history) {
      auto it = tmp.find(el.timestamp);
      if (it != tmp.end()) {
        REF_ASSERT(el.rollback, "event_sink.rs:88");
        tmp.erase(it);
      } else {
        REF_ASSERT(!el.rollback, "event_sink.rs:91");
        tmp.emplace(el.timestamp, el);
      }
    }
    events->clear();
    l1_messages->clear();
    for (const auto& kv : tmp) {
      const LogQuery& el = kv.second;
      EventMessage m{el.shard_id, el.is_service, el.tx_number_in_block, el.address, el.key, el.written_value};
      (el.aux_byte == event_aux ? *events : *l1_messages).push_back(m);  // :124-128
    }
  }
  void start_frame(uint32_t) { frames_stack.emplace_back(); }
  void finish_frame(bool panicked, uint32_t) {
    REF_ASSERT(!frames_stack.empty(), "frame must be started before finishing");
    ApplicationData cur = std::move(frames_stack.back());
    frames_stack.pop_back();
    REF_ASSERT(!frames_stack.empty(), "parent_frame_must_exist");
    ApplicationData& parent = frames_stack.back();
    parent.forward.insert(parent.forward.end(), cur.forward.begin(), cur.forward.end());
    if (panicked)
      parent.forward.insert(parent.forward.end(), cur.rollbacks.rbegin(), cur.rollbacks.rend());
    else
      parent.rollbacks.insert(parent.rollbacks.end(), cur.rollbacks.begin(), cur.rollbacks.end());
  }
};

// ---------------------------------------------------------------------------------------
// SimpleDecommitter — reference_impls/decommitter.rs:10-99
// ---------------------------------------------------------------------------------------
struct U256Hash {
  size_t operator()(const U256& k) const {
    uint64_t h = k.l[0] * 0x9e3779b97f4a7c15ULL;
    h ^= k.l[1] + 0x7f4a7c15ULL + (h << 6) + (h >> 2);
    h ^= k.l[2] + 0x94d049bb133111ebULL + (h << 6) + (h >> 2);
    h ^= k.l[3] + 0xbf58476d1ce4e5b9ULL + (h << 6) + (h >> 2);
    return (size_t)h;
  }
};
struct KnownCode {
  uint32_t blob_id;
  uint32_t preimage_index;  // insertion order of populate (bookkeeping of the build's commitment spec, not reference state)
  CodeBlob words;
};
struct SimpleDecommitter {
  const std::unordered_map<U256, KnownCode, U256Hash>* known_hashes = nullptr;  // :11 (shared, read-only)
  struct Hist {
    uint32_t page;
    uint16_t len;
    uint32_t blob_id;
    uint32_t preimage_index;
  };
  std::unordered_map<U256, Hist, U256Hash> history;  // :12
  // :32-98; returns the blob id of the code for the recorder
  DecommittmentQuery decommit_into_memory(uint32_t cc, DecommittmentQuery q, SimpleMemory& memory, uint32_t* blob_id, uint32_t* preimage_index) {
    auto h = history.find(q.hash);
    if (h != history.end()) {
      q.is_fresh = false;
      q.memory_page = h->second.page;
      q.decommitted_length = h->second.len;
      *blob_id = h->second.blob_id;
      *preimage_index = h->second.preimage_index;
      return q;
    }
    if (!known_hashes) throw RefErr("Code hash must be known");
    auto k = known_hashes->find(q.hash);
    if (k == known_hashes->end()) throw RefErr("Code hash must be known");  // :54-56
    const std::vector<U256>& values = *k->second.words;
    q.decommitted_length = (uint16_t)values.size();
    q.is_fresh = true;
    history[q.hash] = Hist{q.memory_page, q.decommitted_length, k->second.blob_id, k->second.preimage_index};
    MemoryQuery tmp{q.timestamp, MemoryLocation{ZKW_MEM_CODE, q.memory_page, 0}, U256::zero(), false, true};
    for (size_t i = 0; i < values.size(); i++) {
      tmp.location.index = (uint32_t)i;
      tmp.value = values[i];
      memory.specialized_code_query(cc, tmp);
    }
    *blob_id = k->second.blob_id;
    *preimage_index = k->second.preimage_index;
    return q;
  }
};

// ---------------------------------------------------------------------------------------
// VM
// ---------------------------------------------------------------------------------------

struct Decoded {  // zkevm_opcode_defs::DecodedOpcode
  zkw_isa_entry variant;
  uint8_t condition;
  uint8_t src0_reg_idx, src1_reg_idx, dst0_reg_idx, dst1_reg_idx;
  uint16_t imm_0, imm_1;
};

struct PreState {  // cycle.rs:8-14
  PrimitiveValue src0, src1;
  bool has_dst0_mem;
  MemoryLocation dst0_mem_location;
  uint16_t new_pc;
  bool is_kernel_mode;
};

struct Vm {
  const zkw_isa_table* isa = nullptr;
  VmLocalState local_state;
  BlockProperties block_properties;
  InMemoryStorage storage;
  SimpleMemory memory;
  InMemoryEventSink event_sink;
  SimpleDecommitter decommittment_processor;
  Recorder witness_tracer;

  // ---- mod.rs:214-240 ----
  uint32_t timestamp_for_code_or_src_read() const { return local_state.timestamp + 0; }
  uint32_t timestamp_for_first_decommit_or_precompile_read() const { return local_state.timestamp + 1; }
  uint32_t timestamp_for_second_decommit_or_precompile_write() const { return local_state.timestamp + 2; }
  uint32_t timestamp_for_dst_write() const { return local_state.timestamp + 3; }

  uint16_t clip16(uint64_t v) const {  // AllowedPcOrImm::from_u64_clipped (Appendix B hazard)
    if (isa->consts.clip_mode == 0) return v > 0xffff ? 0xffff : (uint16_t)v;
    return (uint16_t)v;
  }
  CallStackEntry& cur() { return local_state.callstack.current; }

  // helpers.rs:318-334
  PrimitiveValue select_register_value(uint8_t idx) const { return idx == 0 ? PrimitiveValue::empty() : local_state.registers[idx - 1]; }
  void update_register_value(uint8_t idx, const PrimitiveValue& v) {
    if (idx > 0) local_state.registers[idx - 1] = v;
  }
  void set_shorthand_panic() { local_state.pending_exception = true; }  // helpers.rs:336-338

  // helpers.rs:10-40 (+78-85)
  MemoryQuery read_code(uint32_t cc, uint32_t ts, MemoryLocation loc) {
    MemoryQuery pq{ts, loc, U256::zero(), false, false};
    MemoryQuery q = memory.read_code_query(cc, pq);
    witness_tracer.add_memory_query(q, 0, cc);
    return q;
  }
  // helpers.rs:53-76
  MemoryQuery read_memory(uint32_t cc, uint32_t ts, MemoryLocation loc) {
    MemoryQuery pq{ts, loc, U256::zero(), false, false};
    MemoryQuery q = memory.execute_partial_query(cc, pq);
    witness_tracer.add_memory_query(q, 0, cc);
    return q;
  }
  // helpers.rs:87-117
  MemoryQuery write_memory(uint32_t cc, uint32_t ts, MemoryLocation loc, const PrimitiveValue& v) {
    MemoryQuery pq{ts, loc, v.value, v.is_pointer, true};
    MemoryQuery q = memory.execute_partial_query(cc, pq);
    witness_tracer.add_memory_query(q, 0, cc);
    return q;
  }
  // helpers.rs:119-136
  uint32_t refund_for_partial_query(uint32_t cc, const LogQuery& pq) {
    REF_ASSERT(pq.rw_flag == true, "refund for read");
    uint32_t refund = storage.estimate_refunds_for_write(cc, pq);
    witness_tracer.add_log(pq, ZKW_LQ_REFUND, cc);
    return refund;
  }
  // helpers.rs:138-155
  LogQuery access_storage(uint32_t cc, LogQuery query) {
    query = storage.execute_partial_query(cc, query);
    if (!query.rw_flag) query.written_value = query.read_value;
    witness_tracer.add_log(query, ZKW_LQ_LOG, cc);
    return query;
  }
  // helpers.rs:157-162
  void emit_event(uint32_t cc, const LogQuery& query) {
    event_sink.add_partial_query(cc, query, isa->consts.event_aux_byte, isa->consts.l1_message_aux_byte);
    if (witness_tracer.cb) witness_tracer.cb->log(cblog::EV_ADD_PARTIAL_QUERY, cc, Recorder::cb_log(query));
    witness_tracer.add_log(query, ZKW_LQ_LOG, cc);
  }
  // helpers.rs:164-194
  DecommittmentQuery decommit(uint32_t cc, const U256& hash, uint32_t candidate_page, uint32_t ts) {
    DecommittmentQuery pq{hash, ts, candidate_page, 0, false};
    uint32_t blob_id = 0, preimage_index = 0;
    DecommittmentQuery q = decommittment_processor.decommit_into_memory(cc, pq, memory, &blob_id, &preimage_index);
    const std::vector<U256>* words = nullptr;
    if (q.is_fresh) words = decommittment_processor.known_hashes->find(q.hash)->second.words.get();
    witness_tracer.decommit(q, blob_id, preimage_index, cc, words);
    return q;
  }
  void call_precompile(uint32_t cc, const LogQuery& query);  // helpers.rs:196-223
  // helpers.rs:225-246
  void start_frame(uint32_t cc, const CallStackEntry& context_entry) {
    uint32_t ts = local_state.timestamp;
    storage.start_frame(ts);
    event_sink.start_frame(ts);
    if (witness_tracer.cb) witness_tracer.cb->simple(cblog::EV_START_FRAME, ts, 0);
    witness_tracer.frame_start(local_state.callstack.current, context_entry, !context_entry.is_local_frame, cc);
    local_state.callstack.push_entry(context_entry);
  }
  // helpers.rs:248-264
  CallStackEntry finish_frame(uint32_t cc, bool panicked) {
    uint32_t ts = local_state.timestamp;
    storage.finish_frame(ts, panicked);
    event_sink.finish_frame(panicked, ts);
    if (witness_tracer.cb) witness_tracer.cb->simple(cblog::EV_FINISH_FRAME, panicked, ts);
    witness_tracer.frame_finish(panicked, cc);
    return local_state.callstack.pop_entry();
  }
  // helpers.rs:266-283
  void perform_dst0_update(uint32_t cc, const PrimitiveValue& value, const PreState& ps, const Decoded& op) {
    if (ps.has_dst0_mem)
      write_memory(cc, timestamp_for_dst_write(), ps.dst0_mem_location, value);
    else
      update_register_value(op.dst0_reg_idx, value);
  }
  void perform_dst1_update(const PrimitiveValue& value, uint8_t idx) { update_register_value(idx, value); }  // :285-287

  bool compute_addresses(uint16_t& sp, uint8_t reg_idx, uint16_t imm, uint8_t mode, bool is_write, PrimitiveValue* reg_value, MemoryLocation* loc);
  Decoded read_and_decode(bool* skip_cycle);
  void cycle();
  void apply(const Decoded& op, const PreState& ps);
  void add_sub(const Decoded& op, const PreState& ps, bool is_sub);
  void mul(const Decoded& op, const PreState& ps);
  void div(const Decoded& op, const PreState& ps);
  void shift(const Decoded& op, const PreState& ps);
  void binop(const Decoded& op, const PreState& ps);
  void ptr(const Decoded& op, const PreState& ps);
  void context(const Decoded& op, const PreState& ps);
  void near_call(const Decoded& op, const PreState& ps);
  void log(const Decoded& op, const PreState& ps);
  void far_call(const Decoded& op, const PreState& ps);
  void ret(const Decoded& op, const PreState& ps);
  void uma(const Decoded& op, const PreState& ps);
};

// precompiles (zk_evm_abstractions::precompiles, absent crate — see hashes.hpp header)
// `rounds`: the round witness — (reads consumed, writes performed) of every round, round 0 carries the request
void keccak256_rounds_function(uint32_t cc, const LogQuery& params, SimpleMemory& memory, std::vector<MemoryQuery>& reads, std::vector<MemoryQuery>& writes,
                               std::vector<std::pair<uint32_t, uint32_t>>& rounds);
void sha256_rounds_function(uint32_t cc, const LogQuery& params, SimpleMemory& memory, std::vector<MemoryQuery>& reads, std::vector<MemoryQuery>& writes,
                            std::vector<std::pair<uint32_t, uint32_t>>& rounds);
void ecrecover_function(uint32_t cc, const LogQuery& params, SimpleMemory& memory, uint32_t layout, std::vector<MemoryQuery>& reads,
                        std::vector<MemoryQuery>& writes, std::vector<std::pair<uint32_t, uint32_t>>& rounds);

}  // namespace zko
