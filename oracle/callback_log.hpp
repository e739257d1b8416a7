// ORACLE — TEST INFRASTRUCTURE ONLY.
// A canonical, hashable log of the reference's outward calls in call order — the 10 VmWitnessTracer
// callbacks (witness_trace/mod.rs:11-72) and the 3 EventSink calls (event_sink.rs:134-176).  The
// oracle fills it directly while it runs the restated cycle(); tests/host/replay_lib.cpp fills it
// from the C++ host mirror (era-zk_evm_amd/host/zk_evm.hpp) replaying a GPU trace; the two logs
// must be identical entry for entry.  One FNV-1a-64 value per call over (callback id, cc, payload).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/zkw.h"

namespace cblog {

enum { START_CYCLE = 1, END_CYCLE, ADD_MEMORY_QUERY, RECORD_REFUND, ADD_LOG_QUERY, ADD_DECOMMITTMENT, ADD_PRECOMPILE_CALL_RESULT, START_CONTEXT,
       FINISH_CONTEXT, EV_ADD_PARTIAL_QUERY, EV_START_FRAME, EV_FINISH_FRAME };

struct Hasher {
  uint64_t h = 0xcbf29ce484222325ULL;
  void bytes(const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ULL;
  }
  void u32(uint32_t v) { bytes(&v, 4); }
  void u64(uint64_t v) { bytes(&v, 8); }
};

struct MemQ {
  uint32_t timestamp, page, index;
  uint8_t type, is_ptr, rw;
  uint64_t value[4];
};
struct LogQ {
  uint32_t timestamp;
  uint16_t tx;
  uint8_t aux, shard, rw, rollback, is_service;
  uint8_t address[20];
  uint64_t key[4], read[4], written[4];
};

struct Log {
  std::vector<uint64_t> entries;
  static void put(Hasher& h, const MemQ& q) {
    h.u32(q.timestamp); h.u32(q.page); h.u32(q.index); h.u32(q.type | (q.is_ptr << 8) | (q.rw << 16)); h.bytes(q.value, 32);
  }
  static void put(Hasher& h, const LogQ& q) {
    h.u32(q.timestamp); h.u32(q.tx); h.u32(q.aux | (q.shard << 8) | (q.rw << 16) | (q.rollback << 17) | (q.is_service << 18));
    h.bytes(q.address, 20); h.bytes(q.key, 32); h.bytes(q.read, 32); h.bytes(q.written, 32);
  }
  static void put_entry(Hasher& h, const zkw_callstack_entry& e) {
    zkw_callstack_entry c = e;
    c.reserved0 = 0;
    c.reserved1 = 0;
    h.bytes(&c, sizeof c);
  }
  void state(uint32_t id, const zkw_vm_local_state& st, const zkw_callstack_entry* inner) {
    Hasher h;
    h.u32(id);
    zkw_vm_local_state c = st;
    c.current.reserved0 = 0;
    c.current.reserved1 = 0;
    h.bytes(&c, sizeof c);
    for (uint32_t d = 0; d < st.callstack_depth; d++) put_entry(h, inner[d]);
    entries.push_back(h.h);
  }
  void mem(uint32_t cc, const MemQ& q) {
    Hasher h; h.u32(ADD_MEMORY_QUERY); h.u32(cc); put(h, q); entries.push_back(h.h);
  }
  void log(uint32_t id, uint32_t cc, const LogQ& q, uint32_t extra = 0) {
    Hasher h; h.u32(id); h.u32(cc); put(h, q); h.u32(extra); entries.push_back(h.h);
  }
  void decommit(uint32_t cc, const uint64_t hash[4], uint32_t ts, uint32_t page, uint32_t len, bool fresh, const uint64_t* words, size_t n_words) {
    Hasher h; h.u32(ADD_DECOMMITTMENT); h.u32(cc); h.bytes(hash, 32); h.u32(ts); h.u32(page); h.u32(len); h.u32(fresh); h.u64(n_words);
    h.bytes(words, n_words * 32);
    entries.push_back(h.h);
  }
  // round witness (PrecompileCyclesWitness, witness_trace/mod.rs:43-50): per round, whether it carries the request,
  // how many of the call's reads it consumed and whether it performed the call's writes
  struct Round {
    uint8_t has_new_request;
    uint32_t n_reads, n_writes;
  };
  void precompile(uint32_t cc, const LogQ& call, const std::vector<MemQ>& in, const std::vector<MemQ>& out, uint32_t kind, const std::vector<Round>& rounds) {
    Hasher h; h.u32(ADD_PRECOMPILE_CALL_RESULT); h.u32(cc); put(h, call); h.u64(in.size());
    for (auto& q : in) put(h, q);
    h.u64(out.size());
    for (auto& q : out) put(h, q);
    h.u32(kind); h.u64(rounds.size());
    for (auto& r : rounds) { h.u32(r.has_new_request); h.u32(r.n_reads); h.u32(r.n_writes); }
    entries.push_back(h.h);
  }
  void frame_start(uint32_t cc, const zkw_callstack_entry& prev, const zkw_callstack_entry& next) {
    Hasher h; h.u32(START_CONTEXT); h.u32(cc); put_entry(h, prev); put_entry(h, next); entries.push_back(h.h);
  }
  void simple(uint32_t id, uint32_t a, uint32_t b) {
    Hasher h; h.u32(id); h.u32(a); h.u32(b); entries.push_back(h.h);
  }
};

}  // namespace cblog
