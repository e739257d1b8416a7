// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the build's OWN queue-commitment spec "ZKW-GL-sponge v2" (DESIGN.md
// §commitments).  The reference has no sponge / queue commitment at all, so this
// checks the HIP kernels (era-zk_evm_amd/csrc/zkw_commit.hip) against an independently written
// implementation of the same spec — arithmetic here is plain `unsigned __int128 % p`.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/zkw.h"

namespace zko {
namespace gl {

static const uint64_t P = 0xffffffff00000001ULL;
inline uint64_t add(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }
inline uint64_t mul(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % P); }
inline uint64_t pow7(uint64_t x) {
  uint64_t x2 = mul(x, x), x4 = mul(x2, x2), x6 = mul(x4, x2);
  return mul(x6, x);
}

struct Perm {
  uint64_t rc[4 * 12 + 22 + 4 * 12];
  Perm() {  // splitmix64("zkwGLv1"), rejecting values >= p
    uint64_t x = 0x7a6b77474c7631ULL;
    int n = 0;
    while (n < (int)(sizeof rc / sizeof rc[0])) {
      x += 0x9E3779B97F4A7C15ULL;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
      z = z ^ (z >> 31);
      if (z < P) rc[n++] = z;
    }
  }
  static void external(uint64_t s[12]) {
    static const uint64_t M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
    uint64_t t[12];
    for (int b = 0; b < 3; b++)
      for (int i = 0; i < 4; i++) {
        uint64_t acc = 0;
        for (int j = 0; j < 4; j++) acc = add(acc, mul(M4[i][j], s[4 * b + j]));
        t[4 * b + i] = acc;
      }
    for (int i = 0; i < 12; i++) s[i] = add(t[i], add(add(t[i & 3], t[4 + (i & 3)]), t[8 + (i & 3)]));  // circ(2 M4, M4, M4)
  }
  static void internal(uint64_t s[12]) {
    uint64_t sum = 0;
    for (int i = 0; i < 12; i++) sum = add(sum, s[i]);
    for (int i = 0; i < 12; i++) s[i] = add(sum, mul(s[i], 1ULL << i));  // J + diag(2^i)
  }
  void operator()(uint64_t s[12]) const {
    external(s);
    int k = 0;
    for (int r = 0; r < 4; r++) {
      for (int i = 0; i < 12; i++) s[i] = pow7(add(s[i], rc[k + i]));
      k += 12;
      external(s);
    }
    for (int r = 0; r < 22; r++) {
      s[0] = pow7(add(s[0], rc[k++]));
      internal(s);
    }
    for (int r = 0; r < 4; r++) {
      for (int i = 0; i < 12; i++) s[i] = pow7(add(s[i], rc[k + i]));
      k += 12;
      external(s);
    }
  }
};

struct Digest {
  uint64_t v[4];
};

inline Digest leaf(const Perm& perm, uint32_t type, const std::vector<uint64_t>& f) {
  uint64_t s[12] = {0};
  s[8] = ((uint64_t)type << 32) | (uint64_t)f.size();
  for (size_t b = 0; b * 8 < f.size(); b++) {
    for (size_t j = 0; j < 8 && b * 8 + j < f.size(); j++) s[j] = add(s[j], f[b * 8 + j] % P);
    perm(s);
  }
  return Digest{{s[0], s[1], s[2], s[3]}};
}
// x10 / x11: the two spare inputs of the chain permutation (decommit queue: the per-record fields; zero elsewhere)
inline void chain_step(const Perm& perm, const Digest& lf, Digest& tail, uint64_t index_plus_1, uint32_t queue_id, uint64_t x10 = 0, uint64_t x11 = 0) {
  uint64_t s[12] = {lf.v[0], lf.v[1], lf.v[2], lf.v[3], tail.v[0], tail.v[1], tail.v[2], tail.v[3], index_plus_1, queue_id, x10 % P, x11 % P};
  perm(s);
  for (int i = 0; i < 4; i++) tail.v[i] = s[i];
}

inline std::vector<uint64_t> limbs(const zkw_u256& v) {
  std::vector<uint64_t> r;
  for (int i = 0; i < 4; i++) {
    r.push_back((uint32_t)v.l[i]);
    r.push_back((uint32_t)(v.l[i] >> 32));
  }
  return r;
}

// two levels: chains inside 64-word chunks (queue id 0xB10B, index restarts at 1 in every chunk), then one chain over
// the chunk tails (queue id 0xB10C)
inline Digest blob_digest(const Perm& perm, const zkw_u256* words, size_t n) {
  Digest top{{0, 0, 0, 0}};
  uint64_t c = 0;
  for (size_t first = 0; first < n; first += 64) {
    Digest tail{{0, 0, 0, 0}};
    const size_t cnt = n - first < 64 ? n - first : 64;
    for (size_t j = 0; j < cnt; j++) chain_step(perm, leaf(perm, 4, limbs(words[first + j])), tail, j + 1, 0xB10B);
    chain_step(perm, tail, top, ++c, 0xB10C);
  }
  return top;
}

// Memory and log queues: the record itself is the input of the chain permutation — no leaf.  A record of 2k u32 words
// w[0..2k) is packed into field elements below 2^56 (a 64-bit value need not be a canonical element, 7 bytes always are):
// element i < k = w[2i] | (w[2i+1] & 0xffffff) << 32, then the k dropped top bytes of the odd words, seven per element.
// Each block of 7 elements is one permutation:  tail' = P(e[7b..7b+7) | tail | (j + 1) | queue << 40 | b << 48)[0..4].
// A memory query (12 words -> 6 + 1 elements) is one permutation, a log query (32 words -> 16 + 3) is three.
inline std::vector<uint64_t> pack56(const std::vector<uint32_t>& w) {
  std::vector<uint64_t> e;
  const size_t k = w.size() / 2;
  for (size_t i = 0; i < k; i++) e.push_back((uint64_t)w[2 * i] | ((uint64_t)(w[2 * i + 1] & 0xffffffu) << 32));
  for (size_t first = 0; first < k; first += 7) {
    uint64_t x = 0;
    for (size_t i = first; i < k && i < first + 7; i++) x |= (uint64_t)(w[2 * i + 1] >> 24) << (8 * (i - first));
    e.push_back(x);
  }
  return e;
}
inline void chain_record(const Perm& perm, const std::vector<uint64_t>& e, Digest& tail, uint64_t index_plus_1, uint32_t queue_id) {
  for (size_t b = 0; b * 7 < e.size(); b++) {
    uint64_t s[12] = {0};
    for (size_t i = 0; i < 7 && b * 7 + i < e.size(); i++) s[i] = e[b * 7 + i];
    for (int i = 0; i < 4; i++) s[7 + i] = tail.v[i];
    s[11] = index_plus_1 | ((uint64_t)queue_id << 40) | ((uint64_t)b << 48);
    perm(s);
    for (int i = 0; i < 4; i++) tail.v[i] = s[i];
  }
}
inline std::vector<uint32_t> words32(const zkw_u256& v) {
  std::vector<uint32_t> r;
  for (int i = 0; i < 4; i++) {
    r.push_back((uint32_t)v.l[i]);
    r.push_back((uint32_t)(v.l[i] >> 32));
  }
  return r;
}

inline Digest mem_queue(const Perm& perm, const zkw_mem_query* q, size_t n) {
  Digest tail{{0, 0, 0, 0}};
  for (size_t j = 0; j < n; j++) {
    std::vector<uint32_t> w = {q[j].timestamp, q[j].page, q[j].index, q[j].meta};
    auto v = words32(q[j].value);
    w.insert(w.end(), v.begin(), v.end());
    chain_record(perm, pack56(w), tail, j + 1, ZKW_QUEUE_MEMORY);
  }
  return tail;
}
inline Digest log_queue(const Perm& perm, const zkw_log_query* q, size_t n) {
  Digest tail{{0, 0, 0, 0}};
  for (size_t j = 0; j < n; j++) {
    std::vector<uint32_t> w = {q[j].timestamp, q[j].tx_number_in_block,
                               (uint32_t)q[j].aux_byte | ((uint32_t)q[j].shard_id << 8) | ((uint32_t)q[j].bools << 16) | ((uint32_t)q[j].kind << 24)};
    for (int a = 0; a < 5; a++) {
      uint32_t x;
      std::memcpy(&x, q[j].address + 4 * a, 4);
      w.push_back(x);
    }
    for (const zkw_u256* x : {&q[j].key, &q[j].read_value, &q[j].written_value}) {
      auto v = words32(*x);
      w.insert(w.end(), v.begin(), v.end());
    }
    chain_record(perm, pack56(w), tail, j + 1, ZKW_QUEUE_LOG);
  }
  return tail;
}
inline Digest decommit_queue(const Perm& perm, const zkw_aux_event* e, size_t n, const std::vector<Digest>& blob_digests) {
  Digest tail{{0, 0, 0, 0}};
  uint64_t j = 0;
  for (size_t k = 0; k < n; k++) {
    if (e[k].type != ZKW_AUX_DECOMMIT) continue;
    // leaf = what identifies the code (8 limbs of the code hash, blob length, blob digest: cacheable per hash); the
    // per-record fields (timestamp | fresh << 32, page) are the spare inputs of the chain step: one permutation per record
    std::vector<uint64_t> f = limbs(e[k].u.hash);
    f.push_back((uint64_t)(e[k].c & 0xffffu));
    const Digest& bd = blob_digests[e[k].c >> 16];
    for (int i = 0; i < 4; i++) f.push_back(bd.v[i]);
    chain_step(perm, leaf(perm, 3, f), tail, ++j, ZKW_QUEUE_DECOMMIT, (uint64_t)e[k].a | ((uint64_t)e[k].flag << 32), (uint64_t)e[k].b);
  }
  return tail;
}

}  // namespace gl
}  // namespace zko
