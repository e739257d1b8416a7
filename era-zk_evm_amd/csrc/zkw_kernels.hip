// EraVM batch witness kernel for MI355X (gfx950).
//
// One VM instance per lane, one wave per 64 instances, 4 waves per workgroup, TWO workgroups per CU (two waves per SIMD:
// the kernel is bound by the latency of one VM cycle of one wave, so the second wave of a SIMD fills the first one's
// stalls).  Every active lane of a wave executes `cycle()` of the reference (src/vm_state/cycle.rs:257-429) once per
// loop iteration, in lockstep:
//   * the 15 x 256-bit register file of each lane lives in the upper half of the lane's vector registers (v128..v255,
//     struct RegFile).  Decode is scalar per group of lanes that hold the same opcode word, so a register number is
//     wave-uniform and a register access is VGPR-indexed addressing (s_set_gpr_idx_on + 8 v_mov): no LDS round trip,
//     and the LDS footprint of a wave drops from 32 KB to 6 KB, which is what lets a second workgroup share the CU;
//   * the cold per-lane scalars (context value, pubdata counters, arena bookkeeping) live in LDS ([field][lane]),
//     the hot ones in VGPRs; the heavy, rare opcode bodies (far_call, ret, near_call, log + precompiles) are out-of-line
//     functions (one per opcode behind a tail-calling dispatcher: one call site) so that their register demand does not
//     add to the 120 + ~128 registers of the hot loop;
//   * lanes of a wave that run different programs are grouped by decoded variant (same ISA entry, per-lane register
//     numbers / immediates): such a group runs out of line with waterfall register access (zkw_vec_exec);
//   * the opcode stream is fetched from HBM as 32-byte code words (4 opcodes), cached in LDS exactly like
//     `previous_code_word` (cycle.rs:53-101);
//   * the packed ISA table (host-uploaded, 2048 x 8 B) is staged in LDS once per workgroup;
//   * stack / heap / aux-heap pages are interleaved across the lanes of a wave ([word][2][lane]) so a shared tape gives
//     fully coalesced accesses; pages are lazily zeroed through a per-page high-water mark instead of memsets;
//   * the sparse per-cycle query logs (memory / log / aux) are compacted per wave with ballot + popcount prefix sums
//     into dense streams (lane- and sequence-tagged records);
//   * the CycleRecord goes out in delta form: a 32-byte tail per lane and cycle plus the values of the registers the
//     cycle wrote, compacted per wave.
// No MFMA: there is no dense contraction on this path (256-bit integer ALU, byte shuffles, hashes).
//
// Reference citations (file:line) are relative to /root/reference/src.
#include <hip/hip_runtime.h>

#include <mutex>
#include <type_traits>

#include "zkw_device.h"
#include "zkw_u256.hip.h"
#include "zkw_goldilocks.hip.h"

// Divergence annotations.  A region that only some lanes of a wave enter AND that holds a cross-lane operation (a ballot, a
// stream allocation, a readlane ...) is opened with ZKW_DIV_IF instead of `if`; a loop or function body that lanes leave
// at different times and that holds one is wrapped in ZKW_DIV_SCOPE.  On the device both are nothing (`if` / empty: the
// compiler's structurizer and the execution mask do the work); the 64-lane CPU emulation of tests/emu — every lane a
// fiber — takes its execution mask from them (tests/emu/emu_simt.cpp) and aborts on a cross-lane operation that not
// every active lane reaches, so a missing annotation is an error there, never a silent difference.
// (defined in zkw_device.h; ZKW_LOCKSTEP() marks a place where the code relies on the lanes of a wave running in lockstep
// through wave-shared memory: nothing on the device, a rendezvous of the lanes in the emulation.)

// The HOST pass of the device build parses the device functions below and never runs them: stand-ins for what only the
// emulation build's hip_runtime.h provides to the non-device branches (the wave's cursor registers, the yield of a waiting wave).
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(ZKW_EMU_BUILD)
static __host__ __device__ inline uint32_t* zkw_emu_wave_sregs() { return nullptr; }
#define ZKW_EMU_FETCH_ADD(reg, n) (zkw_emu_wave_sregs()[(reg)] + (n))
#define ZKW_EMU_YIELD() ((void)0)
#endif
#ifndef ZKW_EMU_BUILD
extern __shared__ uint4 zkw_lds[];  // the dynamic LDS segment of the cycle kernel's workgroups (layout: ZKW_LDS_WAVES0 below)
#endif  // (the emulation build's stand-in header declares the segment)

// Test hook of the CPU emulation builds (tests/emu): lane-cycles by path, so that a test can assert that a tape really
// went through the short cycle / a variant group.  Nothing in the product.
#ifdef ZKW_EMU_BUILD
extern "C" unsigned long long zkw_emu_path_counts[8];  // [0] short cycle, [1] ... of them heap / aux accesses, [2] general path, [3] ... of them in a variant group, [4] keccak256 calls served by helper waves, [5] decommits chained by helper waves
#define ZKW_EMU_COUNT(i) (zkw_emu_path_counts[(i)]++)
#else
#define ZKW_EMU_COUNT(i) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// per-lane execution context (lives in VGPRs; every helper below is force-inlined)
// ---------------------------------------------------------------------------------------------
struct Lane {
  // Lane of the wave.  Re-derived from the hardware (zkw_lane_id) at the top of every cycle and of every opcode body
  // instead of being kept from the kernel start: the addresses derived from it then cannot be hoisted out of the cycle
  // loop (some forty of them were, each owning a register for the whole loop, all of them in scratch memory), and the
  // handful of asm statements per cycle this costs does not fence the scheduler the way one per access would.
  u32 lane;
  // what every cycle touches stays in vector registers (12 more)
  u32 pc, sp, ergs, timestamp, prev_super_pc, depth, status;
  u32 flags;       // FLAG_*
  u32 kflags;      // KF_*: frame properties + per-cycle markers
  u32 ptr_bitmap;  // bit r = register r + 1 holds a pointer
  u32 reg_dirty;   // registers written in this cycle (bit r = register r + 1)
  u32 counts;      // per-cycle, saturating bytes: in-cycle sequence number | memory queries << 8 | log queries << 16 | aux events << 24
};
#define KF_KERNEL 1u       /* callstack.current.is_kernel_mode() */
#define KF_STATIC 2u       /* callstack.current.is_static */
#define KF_LOCAL 4u        /* callstack.current.is_local_frame */
#define KF_COLD_DIRTY 8u   /* a cold VmLocalState field changed in this cycle (ZKW_AUX_COLD_STATE goes out) */
#define KF_CHARGED 32u     /* this cycle's price is taken and its exceptions / condition are resolved (group loop) */
#define KF_MASKED 64u      /* the instruction of this cycle is the one in sh.enc (pending exception, masked into nop / panic), not the slot of the code word */
#define KF_CODE_PAGE_CHANGED 16u /* previous_code_memory_page != callstack.current.code_page (cycle.rs:49,59) */
#define KF_TAIL2 256u     /* heap bound / aux-heap bound / callstack depth changed since the last record that carried them (CycleRecord, delta form) */
#define KF_DQ_CHAINED 128u /* this cycle chained a decommit into the running commitment (op_far_call): undone if the cycle fails afterwards */
// The rest of the per-lane state lives in LDS, [field][lane] (conflict-free dword accesses); CF(sh, s, field) is an
// lvalue.  Rare opcodes touch the first block, memory operands / frame changes the second.
enum {
  CF_CTX0 = 0,  // context_u128_register, 4 dwords
  CF_SPENT_PUBDATA = 4, CF_MPC, CF_ERGS_PP, CF_TX_NUMBER,
  CF_FIRST_DYN, CF_N_INITIAL_SLOTS, CF_NEXT_SLOT, CF_JOURNAL_LEN, CF_N_HISTORY,
  // hot fields of callstack.current (execution_stack.rs:6-24) and of its memory arena slot
  CF_BASE_PAGE, CF_CODE_PAGE, CF_PREV_CODE_PAGE /* valid while KF_CODE_PAGE_CHANGED */, CF_HEAP_BOUND, CF_AUX_BOUND, CF_CODE_OFF, CF_CODE_LEN, CF_SLOT,
  CF_STACK_HWM, CF_HEAP_HWM, CF_AUX_HWM,
  // run bookkeeping as of the kernel start (the kernel derives the final values from its cycle count)
  CF_N_CYCLES0, CF_CYCLE_COUNTER0,
  ZKW_COLD_FIELDS = 28
};
// LDS rows have a compile-time lane stride (thin waves leave the rest of a row unused), so that a field access is one
// base register + an immediate offset
#ifndef ZKW_EMU_BUILD
#define ZKW_LDS_STRIDE ZKW_WAVE /* the same in the host pass: zkw_cycle_kernel_lds_bytes sizes the launch with it */
#else
#define ZKW_LDS_STRIDE ZKW_EMU_WAVE /* CPU emulation build (tests/emu): one-lane waves, or 64 on the SIMT engine */
#endif
// (the lane index goes through zkw_opaque at every access: a shared, long-lived address register would be the first
// thing the allocator spills around the opcode switch — one v_lshl_add per access is cheaper than that reload)
#if defined(__HIP_DEVICE_COMPILE__)
// On the device the per-lane LDS addresses of a wave live in registers of the reserved range the compiled code never
// touches (see RegFile): v131 = LDS byte address of this lane's column of the cold fields, v132 = of its column of
// the 16-byte slots (pre-decoded code word, masked instruction), v133 = the lane index.  An access copies the
// address out with one volatile v_mov (volatile: as an ordinary value it would be hoisted to one kernel-long register
// and spilled); before, every access recomputed the lane with two v_mbcnt and added a wave base that the allocator
// kept in a spilled scalar register (v_readlane at 41 sites).
#define CF(sh, s, f) (*(ZKW_LDS_AS u32*)(zkw_cold_addr() + (u32)(f) * (ZKW_LDS_STRIDE * 4u)))
#define ZKW_XFER(sh, s, i) (*(ZKW_LDS_AS u32*)(zkw_cold_addr() + ZKW_XFER_OFF + (u32)(i) * (ZKW_LDS_STRIDE * 4u)))
#else
#define CF(sh, s, f) ((sh).cold[(u32)(f) * ZKW_LDS_STRIDE + (s).lane])
#define ZKW_XFER(sh, s, i) ((sh).xfer[(u32)(i) * ZKW_LDS_STRIDE + (s).lane])
#endif
#define lane_inst(sh, s) ((sh).wave * (sh).L + (s).lane)

// The four cold fields that almost every cycle reads — the heap / aux-heap bounds go into every record tail, the arena
// slot and the heap mark into every UMA — live in registers of the reserved range instead of LDS (v129, v130, v134, v135:
// never touched by compiled code, see RegFile): a read is one v_mov instead of an LDS round trip with its s_waitcnt.
// The out-of-line opcode bodies reach them the same way (they are registers of the wave, not of a function).
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKW_CFV(name, reg)                                                        \
  ZD u32 cfv_##name(const Shared&, const Lane&) {                                  \
    u32 x;                                                                         \
    asm volatile("v_mov_b32 %0, " reg : "=v"(x));                                  \
    return x;                                                                      \
  }                                                                                \
  ZD void cfv_set_##name(const Shared&, const Lane&, u32 v) { asm volatile("v_mov_b32 " reg ", %0" : : "v"(v) : reg); }
#endif

// Profiling ablations (debug_flags 1 / 2 / 8 / 32 / 64 / 128, ZKW_NO_PREFETCH: stores or whole phases left out, WRONG results)
// exist only in builds with -DZKW_ABLATION (profiles/tools/r06_ablate.sh builds one): in the product every one of them was
// a scalar test + branch on the hot path — a dozen per VM cycle.  The TEST hooks (4: one lane per group, 1 << 24: variant
// groups forced, 8..23: opcodes kept out of variant groups) stay: they sit on the divergent-wave path only.
#ifdef ZKW_ABLATION
#define ZKW_ABL(flags, bit) (((flags) & (bit)) != 0)
#else
#define ZKW_ABL(flags, bit) false
#endif
// branch layout hints: the common case of the hot path falls through (a taken branch restarts the instruction fetch)
#define ZKW_LIKELY(x) __builtin_expect(!!(x), 1)
#define ZKW_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define FLAG_LT 1u
#define FLAG_EQ 2u
#define FLAG_GT 4u
#define FLAG_PENDING 8u

// dword offsets inside zkw_callstack_entry / zkw_dev_entry
#define E_THIS 0
#define E_SENDER 5
#define E_CODE_ADDR 10
#define E_BASE_PAGE 15
#define E_CODE_PAGE 16
#define E_SP_PC 17
#define E_EH_FLAGS 18
#define E_ERGS 19
#define E_SHARDS 20
#define E_CTX 22
#define E_HEAP_BOUND 26
#define E_AUX_BOUND 27
#define E_CODE_BLOB 28
#define E_SLOT 29
#define E_JOURNAL_MARK 30

ZD void lane_fail(Lane& s, u32 status) {
  if (s.status == ZKW_STATUS_RUNNING) s.status = status;
}
ZD bool lane_ok(const Lane& s) { return s.status == ZKW_STATUS_RUNNING; }

// Volatile view of an LDS word.  The address space is spelled out: a volatile access through a generic pointer is
// not rewritten by the compiler's address-space inference and becomes a FLAT load/store (sc0 sc1) followed by
// s_waitcnt vmcnt(0), i.e. every cursor access waited for ALL outstanding global loads and stores of the wave.
#ifdef __HIP_DEVICE_COMPILE__
#define ZKW_LDS_WORD(p) ((volatile u32 __attribute__((address_space(3)))*)(p))
#else
#define ZKW_LDS_WORD(p) ((volatile u32*)(p))
#endif
// An LDS location as a 32-bit byte address (the hand-over areas of the helper waves are addressed that way: the address
// travels through LDS words and registers).  The emulation build's "LDS" is the array zkw_lds: the address is the offset.
#ifdef __HIP_DEVICE_COMPILE__
#define ZKW_LDS_ADDR(p) ((u32)(size_t)(__attribute__((address_space(3))) uint4*)(p))
#define ZKW_LDS_AT(addr) ZKW_LDS_WORD((__attribute__((address_space(3))) u32*)(size_t)(addr))
#define ZKW_SLEEP(n) __builtin_amdgcn_s_sleep(n)
#else
#define ZKW_LDS_ADDR(p) ((u32)((const char*)(p) - (const char*)zkw_lds))
#define ZKW_LDS_AT(addr) ZKW_LDS_WORD((char*)zkw_lds + (addr))
#define ZKW_SLEEP(n) ZKW_EMU_YIELD() /* a waiting wave hands the processor to the other waves of the workgroup */
#endif
ZD u32 zkw_lds_get(u32 addr) { return *ZKW_LDS_AT(addr); }
ZD void zkw_lds_put(u32 addr, u32 v) { *ZKW_LDS_AT(addr) = v; }
// the four stream cursors of a wave in one volatile 16-byte LDS read
#ifdef __HIP_DEVICE_COMPILE__
ZD uint4 zkw_lds_read4(const u32* p) {
  typedef unsigned int zkw_lds_v4 __attribute__((ext_vector_type(4)));
  const zkw_lds_v4 v = *(volatile zkw_lds_v4 __attribute__((address_space(3)))*)(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
#else
ZD uint4 zkw_lds_read4(const u32* p) { return make_uint4(ZKW_LDS_WORD(p)[0], ZKW_LDS_WORD(p)[1], ZKW_LDS_WORD(p)[2], ZKW_LDS_WORD(p)[3]); }
#endif

#ifdef __HIP_DEVICE_COMPILE__
ZD void zkw_lds_write4(uint4* p, const uint4 v) {
  typedef unsigned int zkw_lds_v4 __attribute__((ext_vector_type(4)));
  zkw_lds_v4 t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *(volatile zkw_lds_v4 __attribute__((address_space(3)))*)(p) = t;
}
#else
ZD void zkw_lds_write4(uint4* p, const uint4 v) { *p = v; }
#endif

// Phase timing of a VM cycle (profiling build only: -DZKW_PROFILE, profiles/tools/r02_phase.sh): shader clocks between
// marks, accumulated per workgroup wave in LDS by the first active lane; printed by one workgroup at the end.
#ifdef ZKW_PROFILE
__shared__ unsigned long long zp_acc[ZKW_MAX_WAVES_PER_GROUP][80];  // 0-3 phases, 8-23 / 24-39 opcode clocks / counts, 40-63 sub-phases
#define ZKW_SUB_DECL unsigned long long zs_last = __builtin_readcyclecounter();
#define ZKW_SUB(i)                                                                       \
  {                                                                                      \
    const unsigned long long zs_now = __builtin_readcyclecounter();                      \
    if (zkw_rank_below(zkw_ballot(1)) == 0) zp_acc[sh.wib][(i)] += zs_now - zs_last;       \
    zs_last = zs_now;                                                                    \
  }
#define ZKW_PROF_DECL unsigned long long zp_last = __builtin_readcyclecounter();
#define ZKW_PROF(i)                                                                      \
  {                                                                                      \
    const unsigned long long zp_now = __builtin_readcyclecounter();                      \
    if (zkw_rank_below(zkw_ballot(1)) == 0) zp_acc[sh.wib][(i)] += zp_now - zp_last;       \
    zp_last = zp_now;                                                                    \
  }
#define ZKW_PROF_RESET zp_last = __builtin_readcyclecounter();
// time stamps that cross the out-of-line call: the previous stamp lives in LDS slot 63
#define ZKW_STAMP0                                                                        \
  {                                                                                      \
    const unsigned long long zt_now = __builtin_readcyclecounter();                      \
    if (zkw_rank_below(zkw_ballot(1)) == 0) zp_acc[sh.wib][63] = zt_now;                    \
  }
#define ZKW_STAMP(i)                                                                     \
  {                                                                                      \
    const unsigned long long zt_now = __builtin_readcyclecounter();                      \
    if (zkw_rank_below(zkw_ballot(1)) == 0) {                                              \
      zp_acc[sh.wib][(i)] += zt_now - zp_acc[sh.wib][63];                                \
      zp_acc[sh.wib][63] = zt_now;                                                       \
    }                                                                                    \
  }
#else
#define ZKW_PROF_DECL
#define ZKW_PROF(i)
#define ZKW_PROF_RESET
#define ZKW_SUB_DECL
#define ZKW_SUB(i)
#define ZKW_STAMP0
#define ZKW_STAMP(i)
#endif

// ---------------------------------------------------------------------------------------------
// wave-level stream compaction: every lane that reaches this point (possibly under divergence)
// gets a unique, dense slot of the wave's stream: ballot -> rank by popcount of lower lanes.  The cursor lives in
// LDS and belongs to this wave alone, and a wave's LDS operations complete in program order, so the base is a plain
// broadcast read by every participating lane followed by one plain write of the leader — no atomic, no shuffle.
// ---------------------------------------------------------------------------------------------
// wave mask of a per-lane condition: the compare writes the mask directly (HIP's zkw_ballot(int) first materialises 0 / 1 per lane)
#ifdef __HIP_DEVICE_COMPILE__
#define zkw_ballot(p) __builtin_amdgcn_ballot_w64((bool)(p))
#else
#define zkw_ballot(p) __ballot((p) ? 1 : 0)
#endif
// number of set bits of `mask` below this lane (v_mbcnt_lo/hi: no per-lane mask register to keep alive)
#ifdef __HIP_DEVICE_COMPILE__
ZD u32 zkw_rank_below(u64 mask) { return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u)); }
// An opaque copy: what is derived from the result cannot be hoisted out of the enclosing loop.  Left alone, the
// optimiser hoists some forty per-lane address computations (LDS field addresses, lane masks, stream offsets) out of
// the cycle loop; each then owns a vector register for the whole loop, and with 128 registers they all end up in
// scratch memory — a reload (a vector-memory round trip behind the outstanding stream stores) where one v_lshl_add
// would have done.
ZD u32 zkw_opaque(u32 x) {
  asm volatile("" : "+v"(x));
  return x;
}
// lane of the wave from the hardware (two v_mbcnt): nothing to keep in a register (or to reload from scratch)
// (inside the volatile asm: the builtins are pure functions of constants and would be hoisted to one kernel-long value)
ZD u32 zkw_lane_id() {
  u32 x;
  asm volatile("v_mov_b32 %0, v133" : "=v"(x));  // set by zkw_set_lane_regs (was: two v_mbcnt)
  return x;
}
#define ZKW_LDS_AS __attribute__((address_space(3)))
ZD u32 zkw_cold_addr() {
  u32 x;
  asm volatile("v_mov_b32 %0, v131" : "=v"(x));
  return x;
}
ZD u32 zkw_slot_addr() {
  u32 x;
  asm volatile("v_mov_b32 %0, v132" : "=v"(x));
  return x;
}
// cold_lds / slot_lds: LDS byte addresses of the wave's cold-field block / 16-byte-slot block; every lane of the wave
ZD void zkw_set_lane_regs(u32 cold_lds, u32 slot_lds, u32 lane) {
  asm volatile("v_mov_b32 v131, %0\n\tv_mov_b32 v132, %1\n\tv_mov_b32 v133, %2" : : "v"(cold_lds + lane * 4u), "v"(slot_lds + lane * 16u), "v"(lane) : "v131", "v132", "v133");
}
// bit `lane` of a wave mask: a select on the mask itself (no 1 << lane register to keep alive)
ZD bool zkw_lane_bit(u64 mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }  // (the mask itself becomes the branch condition: no per-lane value at all)
#else
ZD bool zkw_lane_bit(u64 mask) { return ((mask >> (threadIdx.x & (ZKW_WAVE - 1))) & 1ull) != 0; }
ZD u32 zkw_rank_below(u64 mask) { return (u32)__popcll(mask & ((1ull << (threadIdx.x & (ZKW_WAVE - 1))) - 1ull)); }
ZD u32 zkw_opaque(u32 x) { return x; }
ZD u32 zkw_lane_id() { return threadIdx.x & (ZKW_WAVE - 1); }
#endif

// the 16-byte per-lane LDS slots of a wave: 0..3 = the pre-decoded opcodes of previous_code_word, 4 = the instruction a
// masked / pending lane executes instead (Shared::pcw / enc)
#define ZKW_XFER_OFF ((ZKW_COLD_FIELDS * 4u + 4u * 16u + 16u) * ZKW_LDS_STRIDE)  /* bytes from the cold block to the transfer slot */
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned int zkw_lds_vec4 __attribute__((ext_vector_type(4)));
ZD uint4 zkw_slot_read_at(u32 byte_off) {  // byte_off = slot * 16 * ZKW_LDS_STRIDE (may be a per-lane value)
  const zkw_lds_vec4 v = *(volatile ZKW_LDS_AS zkw_lds_vec4*)(zkw_slot_addr() + byte_off);
  return make_uint4(v.x, v.y, v.z, v.w);
}
ZD void zkw_slot_write_at(u32 byte_off, const uint4 v) {
  zkw_lds_vec4 t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *(volatile ZKW_LDS_AS zkw_lds_vec4*)(zkw_slot_addr() + byte_off) = t;
}
#define ZKW_SLOT_READ(sh, lane, slot) zkw_slot_read_at((u32)(slot) * (16u * ZKW_LDS_STRIDE))
#define ZKW_SLOT_WRITE(sh, lane, slot, v) zkw_slot_write_at((u32)(slot) * (16u * ZKW_LDS_STRIDE), (v))
#else
#define ZKW_SLOT_READ(sh, lane, slot) ((sh).pcw[(u32)(slot) * ZKW_LDS_STRIDE + (lane)])
#define ZKW_SLOT_WRITE(sh, lane, slot, v) ((sh).pcw[(u32)(slot) * ZKW_LDS_STRIDE + (lane)] = (v))
#endif

// On the device the four cursors of a wave (memory / log / aux stream, register deltas) are lanes 0..3 of v128 — a
// register the compiled code never touches (see RegFile) — read and written with v_readlane / v_writelane, which ignore
// the execution mask: a true wave-level scalar that survives divergent control flow, at the price of three instructions
// instead of an LDS round trip per allocation.
#ifdef __HIP_DEVICE_COMPILE__
template <int WHICH>
ZD u32 zkw_cursor_get() {
  u32 v;
  asm volatile("v_readlane_b32 %0, v128, %1" : "=s"(v) : "n"(WHICH));
  return v;
}
template <int WHICH>
ZD void zkw_cursor_set(u32 v) {
  asm volatile("v_writelane_b32 v128, %0, %1" : : "s"(v), "n"(WHICH));
}
template <int WHICH>
ZD u32 stream_alloc(u32*) {
  const u64 mask = zkw_ballot(1);
  const u32 cnt = (u32)__popcll(mask);
  u32 base, next;
  asm volatile("v_readlane_b32 %0, v128, %2\n\ts_nop 0\n\ts_add_u32 %1, %0, %3\n\ts_nop 0\n\tv_writelane_b32 v128, %1, %2"
               : "=&s"(base), "=&s"(next)
               : "n"(WHICH), "s"(cnt)
               : "scc");
  return base + zkw_rank_below(mask);
}
#else  // CPU emulation build: the cursors are scalar registers of the emulated wave (tests/emu/hip/hip_runtime.h)
template <int WHICH>
ZD u32 zkw_cursor_get() { return zkw_emu_wave_sregs()[WHICH]; }
template <int WHICH>
ZD void zkw_cursor_set(u32 v) { zkw_emu_wave_sregs()[WHICH] = v; }  // (wave-uniform value: every active lane writes the same)
template <int WHICH>
ZD u32 stream_alloc(u32*) {
  const u64 mask = zkw_ballot(1);
  const u32 base = ZKW_EMU_FETCH_ADD(WHICH, (u32)__popcll(mask));  // one read-and-advance for the wave
  return base + zkw_rank_below(mask);
}
#endif

// streaming store of a 16-byte unit of a witness stream: written once, read by a later kernel / the host
#ifdef __HIP_DEVICE_COMPILE__
#ifndef ZKW_GLOBAL_AS
#define ZKW_GLOBAL_AS __attribute__((address_space(1)))
#endif
typedef unsigned int zkw_v4u __attribute__((ext_vector_type(4)));
ZD void zkw_stream_store(uint4* p, const uint4 v) {
  zkw_v4u t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
#ifdef ZKW_NO_NT /* (A/B partner: plain stores measured 4 % / 10 % slower on the driver's / the default command, profiles/r08_ab_log.txt) */
  *(ZKW_GLOBAL_AS zkw_v4u*)p = t;
#else
  __builtin_nontemporal_store(t, (zkw_v4u*)p);
#endif
}
#else
ZD void zkw_stream_store(uint4* p, const uint4 v) { *p = v; }
#endif

// Loads / stores of the working arenas (stack, heap, aux heap, code blobs) with the address space spelled out.  Their
// base pointers pass through ZKW_PIN_SGPR (an opaque asm), after which the compiler no longer knows that they point to
// global memory and emits FLAT instructions — which count on BOTH vmcnt and lgkmcnt, so every LDS wait behind one of
// them (the cold lane state lives in LDS) also waits for it to leave the vector-memory queue.
#ifdef __HIP_DEVICE_COMPILE__
ZD uint4 zkw_gload4(const uint4* p) {
  const zkw_v4u v = *(const ZKW_GLOBAL_AS zkw_v4u*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
ZD void zkw_gstore4(uint4* p, const uint4 v) {
  zkw_v4u t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *(ZKW_GLOBAL_AS zkw_v4u*)p = t;
}
ZD uint8_t zkw_gload1(const uint8_t* p) { return *(const ZKW_GLOBAL_AS uint8_t*)p; }
ZD void zkw_gstore1(uint8_t* p, uint8_t v) { *(ZKW_GLOBAL_AS uint8_t*)p = v; }
#else
ZD uint4 zkw_gload4(const uint4* p) { return *p; }
ZD void zkw_gstore4(uint4* p, const uint4 v) { *p = v; }
ZD uint8_t zkw_gload1(const uint8_t* p) { return *p; }
ZD void zkw_gstore1(uint8_t* p, uint8_t v) { *p = v; }
#endif

// "Every vector-memory load issued so far has landed": s_waitcnt vmcnt(0) as a BUILTIN, which the compiler's wait-count
// insertion sees (an asm statement it does not).  Placed right after a group of loads whose values are needed at once
// anyway.  Without it the pass carries "a load may be pending in v[..]" along one path to the next control-flow join and
// puts an s_waitcnt vmcnt(0) THERE — after the opcode body, at the entry of the next one — where, vmcnt being one
// in-order counter for loads and stores, it waits for the acknowledgement of every stream store the body has just issued
// (the cycle kernel spent 54 % of its wave-cycles in s_waitcnt, SQ_WAIT_ANY; the loads account for a third of that).
#ifdef ZKW_WAITPROF  /* profiling build (profiles/tools/r03_waitprof.sh): clocks a wave spends in each of these waits */
__shared__ unsigned long long zw_acc[ZKW_MAX_WAVES_PER_GROUP][32];  // [site] clocks, [16 + site] count
#endif
#ifdef __HIP_DEVICE_COMPILE__
#ifdef ZKW_WAITPROF
#define ZKW_SETTLE(site)                                                                                   \
  {                                                                                                        \
    const unsigned long long zw_t0 = __builtin_readcyclecounter();                                         \
    __builtin_amdgcn_s_waitcnt(0x0f70);                                                                    \
    const unsigned long long zw_t1 = __builtin_readcyclecounter();                                         \
    if (zkw_rank_below(zkw_ballot(1)) == 0) {                                                                \
      zw_acc[threadIdx.x / ZKW_WAVE][(site)] += zw_t1 - zw_t0;                                             \
      zw_acc[threadIdx.x / ZKW_WAVE][16 + (site)] += 1;                                                    \
    }                                                                                                      \
  }
// the same around an LDS / scalar-memory wait (lgkmcnt(0))
#define ZKW_LGKM_PROBE(site)                                                                               \
  {                                                                                                        \
    const unsigned long long zw_t0 = __builtin_readcyclecounter();                                         \
    __builtin_amdgcn_s_waitcnt(0xc07f);                                                                    \
    const unsigned long long zw_t1 = __builtin_readcyclecounter();                                         \
    if (zkw_rank_below(zkw_ballot(1)) == 0) {                                                                \
      zw_acc[threadIdx.x / ZKW_WAVE][(site)] += zw_t1 - zw_t0;                                             \
      zw_acc[threadIdx.x / ZKW_WAVE][16 + (site)] += 1;                                                    \
    }                                                                                                      \
  }
#else
#define ZKW_SETTLE(site) __builtin_amdgcn_s_waitcnt(0x0f70) /* vmcnt(0), expcnt / lgkmcnt untouched */
#define ZKW_LGKM_PROBE(site)
#endif
#else
#define ZKW_SETTLE(site)
#define ZKW_LGKM_PROBE(site)
#endif

// orders this wave's own LDS stores before later cross-lane LDS reads/atomics (no workgroup barrier involved)
ZD void zkw_wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// LDS view of one wave.  A workgroup holds ZKW_WAVES_PER_GROUP waves that share the 16 KB ISA table; each wave owns
// 16 B of cursors + per lane: 112 B of state fields (CF_*), 64 B of pre-decoded previous_code_word and 16 B for the pending instruction.
// The Keccak row (rare, precompile only) is in HBM.
struct Shared {
  uint2* isa;     // [2048] packed ISA table (shared by the waves of the workgroup)
  u32* cursor;    // [4] stream cursors of this wave
  u32* cold;      // [ZKW_COLD_FIELDS][L] cold per-lane state (CF_*)
  u32* krow;      // [34][L] Keccak rate block assembly rows (global memory)
  uint4* pcw;     // [4][L] previous_code_word as 4 pre-decoded opcode slots: (u64 limb k of the word, packed ISA entry of its opcode), lane-minor
  u32* xfer;      // [8][L] one 256-bit value per lane handed to / returned by the out-of-line opcode bodies (zkw_heavy_entry)
  uint4* enc;     // [L] the instruction a lane is about to execute in this cycle: opcode word (lo, hi) + packed ISA entry (attributes, price)
  uint4 *mem_base, *log_base, *aux_base;  // this wave's rows of the query streams (computed once per launch)
  u32 L;
  u32 debug_flags;
  u32 wib;        // wave in workgroup
  u32 wave;       // wave in batch
  // launch-invariant geometry and arena bases, loaded once and pinned in scalar registers (ZKW_PIN_SGPR): left to
  // itself the compiler re-loads them from the parameter block at every use (s_load + s_waitcnt lgkmcnt(0), which
  // also drains the outstanding LDS reads) because an invariant load is cheaper to rematerialise than to keep
  u32 F, S, H, A, cap_mem;
  // constants the hot path reads every cycle (operand addressing, UMA): as loads from the parameter block at their use
  // they were 3.5 scalar-memory round trips per VM cycle (PMC), each followed by s_waitcnt lgkmcnt(0)
  u32 clip_mode, growth_per_byte, max_deref_low, image_words;
  u32* heap_dirty;
  uint4 *stack_vals, *heap, *aux_heap;
  uint8_t* stack_ptrs;
  const uint4* blob_words;
};
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKW_CFV_IN_LDS) /* (-DZKW_CFV_IN_LDS: the A/B partner) */
ZKW_CFV(heap_bound, "v129")
ZKW_CFV(aux_bound, "v130")
ZKW_CFV(slot, "v134")
ZKW_CFV(heap_hwm, "v135")
#else  // CPU emulation builds (and the A/B partner): the LDS rows
#define ZKW_CFV_EMU(name, field)                                                      \
  ZD u32 cfv_##name(const Shared& sh, const Lane& s) { return CF(sh, s, field); }     \
  ZD void cfv_set_##name(const Shared& sh, const Lane& s, u32 v) { CF(sh, s, field) = v; }
ZKW_CFV_EMU(heap_bound, CF_HEAP_BOUND)
ZKW_CFV_EMU(aux_bound, CF_AUX_BOUND)
ZKW_CFV_EMU(slot, CF_SLOT)
ZKW_CFV_EMU(heap_hwm, CF_HEAP_HWM)
#endif
#ifdef __HIP_DEVICE_COMPILE__
#define ZKW_PIN_SGPR(x) asm volatile("" : "+s"(x))
#else
#define ZKW_PIN_SGPR(x) ((void)0)
#endif
// dynamic LDS of a workgroup, in 16-byte units: ISA table | 256-byte sink of the prefetches (prefetch_page_words; kept below
// 64 KB whatever the workgroup size) | one area per cycle wave | the hand-over areas of the helper waves
#define ZKW_LDS_SINK_UNITS 16u
#define ZKW_LDS_WAVES0 (ZKW_ISA_TABLE_SIZE / 2 + ZKW_LDS_SINK_UNITS)
// 16-byte units of LDS per wave: cursors | parameter-block pointer, debug flags | cold | previous_code_word | pending instruction | transfer value
ZD u32 zkw_wave_lds_units() { return 2u + (ZKW_COLD_FIELDS / 4u) * ZKW_LDS_STRIDE + 4u * ZKW_LDS_STRIDE + ZKW_LDS_STRIDE + 2u * ZKW_LDS_STRIDE; }
// `wib` (wave in workgroup), `wave` and `dbg` must be wave-uniform
ZD void shared_setup(Shared& sh, ZKW_KP P, u32 dbg, u32 wib, u32 wave, bool pin) {
  sh.L = P.L;
  sh.debug_flags = dbg;
  sh.wib = wib;
  sh.wave = wave;
  sh.F = P.F; sh.S = P.S; sh.H = P.H; sh.A = P.A; sh.cap_mem = P.cap_mem;
  sh.clip_mode = P.consts.clip_mode; sh.growth_per_byte = P.consts.memory_growth_ergs_per_byte; sh.max_deref_low = P.consts.max_offset_to_deref_low;
  sh.image_words = P.heap_image_words; sh.heap_dirty = P.heap_dirty;
  // this wave's rows: stack_vals / heap / aux_heap [F][words][2][L] x 16 B, stack_ptrs [F][S][L] x 1 B
  sh.stack_vals = P.stack_vals + (u64)wave * P.F * P.S * 2u * P.L; sh.stack_ptrs = P.stack_ptrs + (u64)wave * P.F * P.S * P.L;
  sh.heap = P.heap + (u64)wave * P.F * P.H * 2u * P.L; sh.aux_heap = P.aux_heap + (u64)wave * P.F * P.A * 2u * P.L; sh.blob_words = P.blob_words;
  if (pin) {
    ZKW_PIN_SGPR(sh.clip_mode); ZKW_PIN_SGPR(sh.growth_per_byte); ZKW_PIN_SGPR(sh.max_deref_low); ZKW_PIN_SGPR(sh.image_words); ZKW_PIN_SGPR(sh.heap_dirty);
    ZKW_PIN_SGPR(sh.L); ZKW_PIN_SGPR(sh.F); ZKW_PIN_SGPR(sh.S); ZKW_PIN_SGPR(sh.H); ZKW_PIN_SGPR(sh.A); ZKW_PIN_SGPR(sh.cap_mem);
    ZKW_PIN_SGPR(sh.stack_vals); ZKW_PIN_SGPR(sh.stack_ptrs); ZKW_PIN_SGPR(sh.heap); ZKW_PIN_SGPR(sh.aux_heap); ZKW_PIN_SGPR(sh.blob_words);
  }
  uint4* wl = zkw_lds + ZKW_LDS_WAVES0 + wib * zkw_wave_lds_units();
  sh.isa = (uint2*)zkw_lds;                                  // 16 KB
  sh.cursor = (u32*)wl;                                      // 16 B
  sh.cold = (u32*)(wl + 2);                                  // ZKW_COLD_FIELDS * stride * 4 B   (wl[1]: see zkw_heavy_entry)
  sh.pcw = wl + 2 + (ZKW_COLD_FIELDS / 4u) * ZKW_LDS_STRIDE;  // 4 * stride * 16 B
  sh.enc = wl + 2 + (ZKW_COLD_FIELDS / 4u) * ZKW_LDS_STRIDE + 4u * ZKW_LDS_STRIDE;  // stride * 16 B
  sh.xfer = (u32*)(sh.enc + ZKW_LDS_STRIDE);                 // 8 * stride * 4 B
  sh.krow = P.krow + (u64)wave * ZKW_KROW_WORDS * P.L;
  sh.mem_base = P.mem_stream + (u64)wave * P.cap_mem * 3;
  sh.log_base = P.log_stream + (u64)wave * P.cap_log * 8;
  sh.aux_base = P.aux_stream + (u64)wave * P.cap_aux * 16;
}

ZD u32 next_seq(Lane& s) {
  const u32 q = s.counts & 255u;
  if (ZKW_LIKELY(q != 255u)) s.counts++;
  return q;
}

// WT.add_memory_query (witness_trace/mod.rs:19) / payload of add_precompile_call_result (:43-50)
ZD void emit_mem(ZKW_KP P, Shared& sh, Lane& s, u32 ts, u32 type, u32 page, u32 index, const u256& value, bool is_ptr, bool rw, u32 kind) {
  s.lane = zkw_lane_id();  // fresh, short-lived lane index (see struct Lane)
  const u32 pos = stream_alloc<0>(sh.cursor);
  const u32 seq = next_seq(s);
  if (ZKW_LIKELY((s.counts & 0xff00u) != 0xff00u)) s.counts += 0x100u;
  if (ZKW_UNLIKELY(pos >= sh.cap_mem)) {
    lane_fail(s, ZKW_STATUS_LIMIT);
    return;
  }
  if (ZKW_ABL(sh.debug_flags, 2u)) return;
  const u32 meta = (type & ZKW_MQ_TYPE_MASK) | (is_ptr ? ZKW_MQ_IS_PTR : 0u) | (rw ? ZKW_MQ_RW : 0u) | (kind << ZKW_MQ_KIND_SHIFT);
  // three planes of 16-byte units (header | value low | value high), each [cap_mem]: every store instruction of the
  // wave then covers whole 64-byte lines.  As 48-byte records (three partial-line stores per record) the stream cost
  // 1.7x its own bytes in HBM writes (profiles/r02_traffic.json) — and the kernel is bound by its stores.
  uint4* dst = sh.mem_base + pos;
  // (ablation builds only — pricing what the stream could leave out: 256 = no value planes for code-word queries, 512 = no header plane)
  if (!ZKW_ABL(sh.debug_flags, 512u)) zkw_stream_store(dst, make_uint4(ts, page, index, s.lane | (seq << 8) | (meta << 16)));
  if (ZKW_ABL(sh.debug_flags, 256u) && type == ZKW_MEM_CODE) return;
  zkw_stream_store(dst + sh.cap_mem, u256_lo4(value));
  zkw_stream_store(dst + 2u * (u64)sh.cap_mem, u256_hi4(value));
}

struct LogQ {  // LogQuery (log.rs:85-97)
  u256 key, read_value, written_value;
  u32 address[5];
  u32 timestamp, tx_number, aux_byte, shard_id;
  bool rw, rollback, is_service;
};

// WT.add_log_query / WT.record_refund_for_query (witness_trace/mod.rs:22-33)
ZD void emit_log(ZKW_KP P, Shared& sh, Lane& s, const LogQ& q, u32 kind) {
  const u32 pos = stream_alloc<1>(sh.cursor);
  const u32 seq = next_seq(s);
  if ((s.counts & 0xff0000u) != 0xff0000u) s.counts += 0x10000u;
  if (pos >= P.cap_log) {
    lane_fail(s, ZKW_STATUS_LIMIT);
    return;
  }
  uint4* dst = sh.log_base + (u64)pos * 8;
  zkw_stream_store(dst + 0, u256_lo4(q.key));
  zkw_stream_store(dst + 1, u256_hi4(q.key));
  zkw_stream_store(dst + 2, u256_lo4(q.read_value));
  zkw_stream_store(dst + 3, u256_hi4(q.read_value));
  zkw_stream_store(dst + 4, u256_lo4(q.written_value));
  zkw_stream_store(dst + 5, u256_hi4(q.written_value));
  zkw_stream_store(dst + 6, make_uint4(q.address[0], q.address[1], q.address[2], q.address[3]));
  const u32 bools = (q.rw ? ZKW_LQ_RW : 0u) | (q.rollback ? ZKW_LQ_ROLLBACK : 0u) | (q.is_service ? ZKW_LQ_IS_SERVICE : 0u);
  zkw_stream_store(dst + 7, make_uint4(q.address[4], q.timestamp, (q.tx_number & 0xffffu) | (q.aux_byte << 16) | (q.shard_id << 24),
                      bools | (kind << 8) | (s.lane << 16) | (seq << 24)));
}

// aux events: header + up to 60 payload dwords
ZD uint4* aux_alloc(ZKW_KP P, Shared& sh, Lane& s, u32 type, u32 flag, u32 a, u32 b, u32 c) {
  const u32 pos = stream_alloc<2>(sh.cursor);
  const u32 seq = next_seq(s);
  if ((s.counts & 0xff000000u) != 0xff000000u) s.counts += 0x1000000u;
  if (pos >= P.cap_aux) {
    lane_fail(s, ZKW_STATUS_LIMIT);
    return nullptr;
  }
  uint4* dst = sh.aux_base + (u64)pos * 16;
  dst[0] = make_uint4(type | (s.lane << 8) | (seq << 16) | (flag << 24), a, b, c);
  return dst;
}

// ---------------------------------------------------------------------------------------------
// register file — select_register_value / update_register_value (helpers.rs:318-334)
//
// The 15 x 256-bit registers of a lane live in the UPPER HALF of the lane's vector register file: register i = 1..15,
// limb k is v[128 + 8 i + k] (r0 is the constant zero and is not stored: v128 carries the wave's stream cursors in its
// lanes 0..3 instead, v129..v135 are free).  The kernel is compiled for 128 vector registers
// (__launch_bounds__(256, 4): the compiler — in the kernel and, through the propagated waves-per-eu attribute, in the
// out-of-line functions — allocates v0..v127 only), so v128..v255 are never touched by compiled code and the kernel
// descriptor ends up with 256 registers = two waves per SIMD.  Decode is scalar per group of lanes that hold the same
// opcode word, so a register number is wave-uniform and an access is VGPR-indexed addressing: s_set_gpr_idx_on with
// the scalar offset 8 i, eight v_mov, s_set_gpr_idx_off — no LDS round trip.  (Compared with the register file in LDS: 32 KB less LDS per wave, which is what lets a second
// workgroup share the CU.)  The accesses are `asm volatile`, so they stay in program order among themselves.
// ---------------------------------------------------------------------------------------------
#ifdef __HIP_DEVICE_COMPILE__
struct RegFile {};
// `reg` = 0..15, wave-uniform
ZD u256 rf_get(const RegFile&, u32 reg) {
  u256 v;
  if (reg == 0) return u256_zero();  // r0 reads as zero; its slot (v128..v135) holds the wave's stream cursors instead
  const u32 off = (u32)__builtin_amdgcn_readfirstlane((int)(reg * 8u));
  asm volatile(
      "s_set_gpr_idx_on %8, gpr_idx(SRC0)\n\t"
      "v_mov_b32 %0, v128\n\tv_mov_b32 %1, v129\n\tv_mov_b32 %2, v130\n\tv_mov_b32 %3, v131\n\t"
      "v_mov_b32 %4, v132\n\tv_mov_b32 %5, v133\n\tv_mov_b32 %6, v134\n\tv_mov_b32 %7, v135\n\t"
      "s_set_gpr_idx_off"
      : "=v"(v.w[0]), "=v"(v.w[1]), "=v"(v.w[2]), "=v"(v.w[3]), "=v"(v.w[4]), "=v"(v.w[5]), "=v"(v.w[6]), "=v"(v.w[7])
      : "s"(off)
      : "m0");
  return v;
}
// `reg` = 1..15, wave-uniform; only the active lanes are written
ZD void rf_set(RegFile&, u32 reg, const u256& v) {
  const u32 off = (u32)__builtin_amdgcn_readfirstlane((int)(reg * 8u));
  asm volatile(
      "s_set_gpr_idx_on %8, gpr_idx(DST)\n\t"
      "v_mov_b32 v128, %0\n\tv_mov_b32 v129, %1\n\tv_mov_b32 v130, %2\n\tv_mov_b32 v131, %3\n\t"
      "v_mov_b32 v132, %4\n\tv_mov_b32 v133, %5\n\tv_mov_b32 v134, %6\n\tv_mov_b32 v135, %7\n\t"
      "s_set_gpr_idx_off"
      :
      : "v"(v.w[0]), "v"(v.w[1]), "v"(v.w[2]), "v"(v.w[3]), "v"(v.w[4]), "v"(v.w[5]), "v"(v.w[6]), "v"(v.w[7]), "s"(off)
      : "m0");
}
// the ends of the reserved register range are named once, so that the kernel descriptor covers it
ZD void rf_init(RegFile&) {
  asm volatile("v_mov_b32 v129, 0\n\tv_mov_b32 v255, 0" : : : "v128", "v129", "v255");  // (v128 holds the stream cursors: not touched here)
}
// The same register file addressed with a PER-LANE register number (variant grouping, zkw_vec_exec): a waterfall — take
// the first remaining lane's number, serve every lane that holds the same one with the scalar-indexed access, repeat.
// One pass per distinct register number among the lanes of the group (at most 16).
struct RegFileVec {};
#define ZKW_RF_IS_VEC(RF) (std::is_same<RF, RegFileVec>::value)
// Written as a wave-uniform loop over the lanes still to serve with the access under a divergent `if` inside it — not as
// the textbook `for (;;) { r = readfirstlane(reg); if (reg == r) { access; break; } }`: there the access sits on the loop's
// exit path, the optimiser moves it behind the loop (same thing for one thread), and behind a divergent loop it runs ONCE,
// for all lanes, with the first lane's register.
ZD u256 rf_get(const RegFileVec&, u32 reg) {
  u256 v = u256_zero();
  u64 todo = zkw_ballot(1);
  while (todo) {
    const u32 r = (u32)__builtin_amdgcn_readlane((int)reg, (int)((u32)__ffsll((long long)todo) - 1u));
    const bool m = reg == r;
    if (m) v = rf_get(RegFile(), r);
    todo &= ~zkw_ballot(m);
  }
  return v;
}
ZD void rf_set(RegFileVec&, u32 reg, const u256& v) {
  u64 todo = zkw_ballot(1);
  while (todo) {
    const u32 r = (u32)__builtin_amdgcn_readlane((int)reg, (int)((u32)__ffsll((long long)todo) - 1u));
    const bool m = reg == r;
    if (m) {
      RegFile f;
      rf_set(f, r, v);
    }
    todo &= ~zkw_ballot(m);
  }
}
#else  // CPU emulation builds of tests/emu: the register file is a struct in the lane's (fiber's) kernel frame
struct RegFile {
  u256 r[16];
};
ZD u256 rf_get(const RegFile& rf, u32 reg) { return rf.r[reg]; }
ZD void rf_set(RegFile& rf, u32 reg, const u256& v) { rf.r[reg] = v; }
ZD void rf_init(RegFile& rf) { rf.r[0] = u256_zero(); }
#ifdef ZKW_WIDE  // the 64-lane emulation forms variant groups: zkw_vec_exec reaches the lane's register file through a pointer
struct RegFileVec {
  RegFile* f;
};
#define ZKW_RF_IS_VEC(RF) (std::is_same<RF, RegFileVec>::value)
ZD u256 rf_get(const RegFileVec& rf, u32 reg) { return rf.f->r[reg & 15u]; }  // (the register number is the lane's own: no waterfall to emulate)
ZD void rf_set(RegFileVec& rf, u32 reg, const u256& v) { rf.f->r[reg & 15u] = v; }
static RegFile* zkw_emu_rf_ptr[ZKW_WAVE * 2 * ZKW_MAX_WAVES_PER_GROUP];  // [thread of the workgroup] -> the register file in its kernel frame
#else
typedef RegFile RegFileVec;  // (the single-lane emulation build never forms a variant group)
#define ZKW_RF_IS_VEC(RF) false
#endif
#endif
template <class RF>
ZD u256 reg_read(Shared& sh, const RF& rf, const Lane& s, u32 idx, u32& is_ptr) {
  is_ptr = ((s.ptr_bitmap << 1) >> idx) & 1u;  // bit idx - 1, nothing for r0 (no `idx - 1` shift count: undefined for a per-lane idx of 0)
  return rf_get(rf, idx);
}
template <class RF>
ZD void reg_write(Shared& sh, RF& rf, Lane& s, u32 idx, const u256& v, bool is_ptr) {
  if (idx == 0) return;
  const u32 r = idx - 1;
  rf_set(rf, idx, v);
  s.reg_dirty |= 1u << r;
  s.ptr_bitmap = (s.ptr_bitmap & ~(1u << r)) | ((is_ptr ? 1u : 0u) << r);
}

// ---------------------------------------------------------------------------------------------
// memory arenas — the device-side SimpleMemory (reference_impls/memory.rs:403-528)
// ---------------------------------------------------------------------------------------------
// A 32-byte word of a lane is stored as two 16-byte halves in two lane-minor planes of its word row
// ([word][2][L] x 16 B: element 2 * w - lane and that + L for the index w returned here), so that each of the two
// load / store instructions of a word access covers whole 64-byte lines.
// `Shared` holds the bases of THIS wave's rows of the arenas (scalar registers); a word is addressed by a 32-bit
// element offset inside the row (the runtime refuses limits whose rows exceed 2^32 elements), so that an access is
// "scalar base + 32-bit vector offset" instead of 64-bit vector arithmetic on a base that has to sit in vector registers.
ZD u32 page_word_index(const Shared& sh, const Lane& s, u32 slot, u32 words_per_page, u32 idx) {
  return (slot * words_per_page + idx) * sh.L + s.lane;
}

// MemoryType::Stack read of the current frame (memory.rs:427-436)
ZD u256 stack_read(ZKW_KP P, const Shared& sh, Lane& s, u32 idx, u32& is_ptr) {
  s.lane = zkw_lane_id();  // fresh, short-lived lane index (see struct Lane)
  is_ptr = 0;
  // a word that was never written reads as zero in the reference (the stack page is a zero-filled Vec); only WRITES
  // need capacity, and stack_hwm <= S
  if (ZKW_UNLIKELY(idx >= CF(sh, s, CF_STACK_HWM))) return u256_zero();
  const u32 w = page_word_index(sh, s, cfv_slot(sh, s), sh.S, idx);
  // (the tag byte last: loads return in order, so a use of the tag scheduled early cannot split the three into two round trips)
  const uint4 lo = zkw_gload4(sh.stack_vals + (2 * w - s.lane)), hi = zkw_gload4(sh.stack_vals + (2 * w - s.lane + sh.L));
#ifdef __HIP_DEVICE_COMPILE__
  asm volatile("" : : : "memory");  // (the scheduler otherwise issues the byte load first and waits for it alone)
#endif
  const uint8_t tag = zkw_gload1(sh.stack_ptrs + w);
  ZKW_SETTLE(1 /* stack operand */);
  is_ptr = tag != 0 ? 1u : 0u;
  return u256_from_uint4(lo, hi);
}
// MemoryType::Stack write (memory.rs:413-425)
ZD void stack_write(ZKW_KP P, const Shared& sh, Lane& s, u32 idx, const u256& v, bool is_ptr) {
  s.lane = zkw_lane_id();  // fresh, short-lived lane index (see struct Lane)
  if (ZKW_UNLIKELY(idx >= sh.S)) {
    lane_fail(s, ZKW_STATUS_LIMIT);
    return;
  }
  for (u32 g = CF(sh, s, CF_STACK_HWM); g < idx; g++) {  // lazily zero the gap
    const u32 w = page_word_index(sh, s, cfv_slot(sh, s), sh.S, g);
    zkw_gstore4(sh.stack_vals + (2 * w - s.lane), make_uint4(0, 0, 0, 0));
    zkw_gstore4(sh.stack_vals + (2 * w - s.lane + sh.L), make_uint4(0, 0, 0, 0));
    zkw_gstore1(sh.stack_ptrs + w, 0);
  }
  const u32 w = page_word_index(sh, s, cfv_slot(sh, s), sh.S, idx);
  if (!ZKW_ABL(sh.debug_flags, 128u)) {  // (128: traffic ablation — the run is then wrong)
    zkw_gstore4(sh.stack_vals + (2 * w - s.lane), u256_lo4(v));
    zkw_gstore4(sh.stack_vals + (2 * w - s.lane + sh.L), u256_hi4(v));
    zkw_gstore1(sh.stack_ptrs + w, is_ptr ? 1 : 0);
  }
  if (idx >= CF(sh, s, CF_STACK_HWM)) CF(sh, s, CF_STACK_HWM) = idx + 1;
}

// heap / aux heap of an arbitrary arena slot; `hwm` is that page's high-water mark
ZD u256 heap_read_at(ZKW_KP P, const Shared& sh, Lane& s, bool is_aux, u32 slot, u32 hwm, u32 idx) {
  s.lane = zkw_lane_id();  // fresh, short-lived lane index (see struct Lane)
  const u32 words = is_aux ? sh.A : sh.H;
  if (ZKW_UNLIKELY(idx >= hwm)) return u256_zero();  // the reference grows its Vec on a read (memory.rs:464,468): not observable; hwm <= words
  const uint4* base = is_aux ? sh.aux_heap : sh.heap;
  const u32 w = page_word_index(sh, s, slot, words, idx);
  return u256_from_uint4(zkw_gload4(base + (2 * w - s.lane)), zkw_gload4(base + (2 * w - s.lane + sh.L)));
}
// MemoryType::Heap / AuxHeap of the current frame (memory.rs:439-473; the page number of the query
// is only debug_assert'ed there, i.e. ignored in release builds)
ZD u256 heap_read_cur(ZKW_KP P, const Shared& sh, Lane& s, bool is_aux, u32 idx) {
  return heap_read_at(P, sh, s, is_aux, cfv_slot(sh, s), is_aux ? CF(sh, s, CF_AUX_HWM) : cfv_heap_hwm(sh, s), idx);
}
// write into page `slot` whose high-water mark is `hwm` (updated; the caller stores it back): the frame fields come in
// as values so that a caller with several accesses reads them from LDS once, together
ZD void heap_write_at(ZKW_KP P, const Shared& sh, Lane& s, bool is_aux, u32 slot, u32& hwm, u32 idx, const u256& v) {
  s.lane = zkw_lane_id();  // fresh, short-lived lane index (see struct Lane)
  const u32 words = is_aux ? sh.A : sh.H;
  if (ZKW_UNLIKELY(idx >= words)) {
    lane_fail(s, ZKW_STATUS_LIMIT);
    return;
  }
  uint4* base = is_aux ? sh.aux_heap : sh.heap;
  for (u32 g = hwm; g < idx; g++) {
    const u32 w = page_word_index(sh, s, slot, words, g);
    zkw_gstore4(base + (2 * w - s.lane), make_uint4(0, 0, 0, 0));
    zkw_gstore4(base + (2 * w - s.lane + sh.L), make_uint4(0, 0, 0, 0));
  }
  const u32 w = page_word_index(sh, s, slot, words, idx);
  if (!ZKW_ABL(sh.debug_flags, 64u)) {  // (64: traffic ablation — the run is then wrong)
    zkw_gstore4(base + (2 * w - s.lane), u256_lo4(v));
    zkw_gstore4(base + (2 * w - s.lane + sh.L), u256_hi4(v));
  }
  if (!is_aux && slot == 0 && idx < sh.image_words && !ZKW_ABL(sh.debug_flags, 32u)) {  // (32: traffic ablation)
    // a word of the uploaded heap image is overwritten: remember it, the next reset restores only those words
    u32* d = sh.heap_dirty + ((u64)sh.wave * ((sh.image_words + 31u) >> 5) + (idx >> 5)) * sh.L + s.lane;
    atomicOr(d, 1u << (idx & 31u));  // result unused: a fire-and-forget atomic instead of a load + store round trip
  }
  if (idx >= hwm) hwm = idx + 1;
}
ZD void heap_write_cur(ZKW_KP P, const Shared& sh, Lane& s, bool is_aux, u32 idx, const u256& v) {
  u32 hwm = is_aux ? CF(sh, s, CF_AUX_HWM) : cfv_heap_hwm(sh, s);
  heap_write_at(P, sh, s, is_aux, cfv_slot(sh, s), hwm, idx, v);
  if (is_aux) CF(sh, s, CF_AUX_HWM) = hwm; else cfv_set_heap_hwm(sh, s, hwm);
}

// Arena slots and page lifetimes (memory.rs:573-758).  A far frame owns one arena slot (stack, heap, aux heap pages
// base + 1 .. 3).  What the reference does with those pages when the frame returns (finish_global_frame, :660-758):
//   * the stack page always goes back to its pool;
//   * the heap / aux page the returndata pointer names moves to `pages_with_extended_lifetime` and stays reachable
//     (Indirection::ReturndataExtendedLifetime) until the frame that RECEIVED it returns — unless that frame forwards
//     the same page as its own returndata, which hands it to the next frame up (:725-742);
//   * every other page of the frame goes back to the pool.
// The state of a slot lives in the `stack_hwm` word of its frame meta (the stack page is dead in every state but LIVE):
//   <= 2^16           LIVE: the stack page's high-water mark
//   ZKW_SLOT_KEPT     | kind << 16 | owner: the heap (kind 2) / aux (kind 3) page is returndata owned by the far frame in
//                     slot `owner`; the slot's other pages read as zero
//   ZKW_SLOT_DEAD     | kind << 16: its owner returned — unreachable for the VM (the reference removes the indirection
//                     and leaks the page: `dump_page_content` still finds it), recycled when no fresh or free slot is left
//   ZKW_SLOT_FREE     returned to the pool: reused by the next far call once the fresh slots are used up
// so that limits.max_far_frames bounds the frames that are live or reachable at one time, not the far calls of a run.
#define ZKW_SLOT_FREE 0xffffffffu
#define ZKW_SLOT_KEPT 0x80000000u
#define ZKW_SLOT_DEAD 0xc0000000u
#define ZKW_SLOT_STATE(st) ((st) & 0xc0000000u)
ZD zkw_dev_frame_meta* frame_metas(ZKW_KP P, const Shared& sh, const Lane& s) { return P.frames + (u64)lane_inst(sh, s) * P.F; }

// MemoryType::FatPointer read (memory.rs:475-521): resolve the page to an arena slot.
// Page 0 is Indirection::Empty; pages that are no heap / aux page of a live or kept frame of this
// instance are "unreachable memory" (the reference's expect() at :478-481).
struct FatPage {  // a resolved page: which arena, which slot, how far it was ever written
  u32 slot, hwm;
  bool found, is_aux, empty;
};
ZD FatPage fat_ptr_resolve(ZKW_KP P, const Shared& sh, Lane& s, u32 page) {
  s.lane = zkw_lane_id();  // fresh, short-lived lane index (see struct Lane)
  FatPage fp;
  fp.slot = 0; fp.hwm = 0; fp.found = false; fp.is_aux = false;
  fp.empty = page == 0;
  if (fp.empty) return fp;
  const u32 cur_slot = cfv_slot(sh, s);
  const u32 rel = page - CF(sh, s, CF_BASE_PAGE);
  if (rel == 2u || rel == 3u) {  // the current frame's own pages: their marks are in registers / LDS
    fp.slot = cur_slot;
    fp.is_aux = rel == 3u;
    fp.hwm = rel == 3u ? CF(sh, s, CF_AUX_HWM) : cfv_heap_hwm(sh, s);
    fp.found = true;
  } else {
    const uint4* fms = (const uint4*)frame_metas(P, sh, s);
    const u32 n = CF(sh, s, CF_NEXT_SLOT);
    for (u32 i = 0; i < n; i++) {
      const uint4 m = fms[i];  // base page, state | stack mark, heap mark, aux mark
      const u32 r = page - m.x;
      if (i == cur_slot || m.x == 0 || m.y == ZKW_SLOT_FREE || (r != 2u && r != 3u)) continue;
      const u32 state = ZKW_SLOT_STATE(m.y);
      if (state == ZKW_SLOT_DEAD) continue;                                   // indirection removed (:752-756)
      if (state == ZKW_SLOT_KEPT && ((m.y >> 16) & 3u) != r) continue;        // the frame's other page went back to the pool
      fp.slot = i;
      fp.is_aux = r == 3u;
      fp.hwm = r == 3u ? m.w : m.z;
      fp.found = true;
    }
  }
  if (!fp.found) lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
  return fp;
}
ZD u256 fat_page_read(const Shared& sh, Lane& s, const FatPage& fp, u32 idx) {
  s.lane = zkw_lane_id();
  if (fp.empty || !fp.found) return u256_zero();
  const u32 words = fp.is_aux ? sh.A : sh.H;
  if (idx >= fp.hwm || idx >= words) return u256_zero();  // `.get(index).unwrap_or(zero)` (:490-495)
  const uint4* base = fp.is_aux ? sh.aux_heap : sh.heap;
  const u32 w = page_word_index(sh, s, fp.slot, words, idx);
  return u256_from_uint4(zkw_gload4(base + (2 * w - s.lane)), zkw_gload4(base + (2 * w - s.lane + sh.L)));
}
ZD u256 fat_ptr_read(ZKW_KP P, const Shared& sh, Lane& s, u32 page, u32 idx) {
  const FatPage fp = fat_ptr_resolve(P, sh, s, page);
  return fat_page_read(sh, s, fp, idx);
}

// read_code_query (memory.rs:556-569) against the blob backing the current code page
ZD u256 code_read(const Shared& sh, const Lane& s, u32 idx) {
  if (idx >= CF(sh, s, CF_CODE_LEN)) return u256_zero();
  const u64 w = (u64)CF(sh, s, CF_CODE_OFF) + idx;
  const uint4 lo = zkw_gload4(sh.blob_words + 2 * w), hi = zkw_gload4(sh.blob_words + 2 * w + 1);
  ZKW_SETTLE(0 /* code word */);
  return u256_from_uint4(lo, hi);
}
// The instruction fetch of a cycle (cycle.rs:76-81).  Instances of a wave usually run the same code at the same pc: the
// word is then ONE scalar-memory load (s_load_dwordx8 through the scalar cache, counted on lgkmcnt) broadcast to the
// lanes, instead of a vector load per lane that queues behind the wave's stream stores (vmcnt is in order: the fetch
// waited ~1000 clocks, 90 times per 256 cycles).  Code blobs are read-only for the lifetime of a batch, so the scalar
// cache cannot hold a stale word.  Lanes that disagree on (blob, length, word index) take the per-lane path.
ZD u256 code_fetch(const Shared& sh, const Lane& s, u32 idx) {
#ifdef ZKW_WIDE
  const u32 c_len = CF(sh, s, CF_CODE_LEN), c_off = CF(sh, s, CF_CODE_OFF);
  const u32 u_len = (u32)__builtin_amdgcn_readfirstlane((int)c_len), u_off = (u32)__builtin_amdgcn_readfirstlane((int)c_off);
  const u32 u_idx = (u32)__builtin_amdgcn_readfirstlane((int)idx);
  if (ZKW_LIKELY(zkw_ballot((c_len != u_len) | (c_off != u_off) | (idx != u_idx)) == 0)) {  // wave-uniform
    u256 v = u256_zero();
    if (ZKW_LIKELY(u_idx < u_len)) {
#ifdef __HIP_DEVICE_COMPILE__
      typedef u32 zkw_v8u __attribute__((ext_vector_type(8)));
      const zkw_v8u w = *(const ZKW_CONST_AS zkw_v8u*)((u64)sh.blob_words + (((u64)u_off + u_idx) << 5));
#else  // (the 64-lane emulation takes the same wave-uniform decision; its "scalar load" is a plain one)
      const u32* w = (const u32*)((const char*)sh.blob_words + (((u64)u_off + u_idx) << 5));
#endif
#pragma unroll
      for (int i = 0; i < 8; i++) v.w[i] = w[i];
    }
    return v;
  }
#endif
  return code_read(sh, s, idx);
}

// ---------------------------------------------------------------------------------------------
// callstack entries in HBM ([inst][depth] x 8 uint4)
// ---------------------------------------------------------------------------------------------
ZD uint4* entry_ptr(ZKW_KP P, const Shared& sh, const Lane& s, u32 depth) { return (uint4*)(P.callstack + ((u64)lane_inst(sh, s) * (P.D + 1) + depth)); }
ZD u32 entry_dword(ZKW_KP P, const Shared& sh, const Lane& s, u32 depth, u32 d) { return ((const u32*)entry_ptr(P, sh, s, depth))[d]; }

// write the hot fields of callstack.current back into its HBM entry
ZD void frame_writeback(ZKW_KP P, const Shared& sh, const Lane& s) {
  u32* e = (u32*)entry_ptr(P, sh, s, s.depth);
  e[E_SP_PC] = (s.sp & 0xffffu) | (s.pc << 16);
  e[E_ERGS] = s.ergs;
  e[E_HEAP_BOUND] = cfv_heap_bound(sh, s);
  e[E_AUX_BOUND] = cfv_aux_bound(sh, s);
}
ZD void hwm_writeback(ZKW_KP P, const Shared& sh, const Lane& s) {
  zkw_dev_frame_meta* fm = P.frames + (u64)lane_inst(sh, s) * P.F + cfv_slot(sh, s);
  fm->stack_hwm = CF(sh, s, CF_STACK_HWM);
  fm->heap_hwm = cfv_heap_hwm(sh, s);
  fm->aux_hwm = CF(sh, s, CF_AUX_HWM);
}
// The hot fields of a callstack entry into the lane.  `e` is the entry image: its row in HBM (a frame the lane returns
// to) or the 32 dwords start_frame has just built in registers (a frame it enters: no reload of what was stored a moment
// ago, which would wait for those stores).  `same_slot`: a near-call frame shares the arena slot, the code blob and the
// page marks of the frame around it — nothing of that is touched.  `fresh_slot`: a far frame that starts now owns empty pages.
template <class E>
ZD void frame_apply(ZKW_KP P, const Shared& sh, Lane& s, const E& e, bool same_slot, bool fresh_slot) {
  s.kflags |= KF_TAIL2;  // depth and memory bounds are those of another frame
  CF(sh, s, CF_BASE_PAGE) = e[E_BASE_PAGE];
  {
    // previous_code_memory_page := the code page of the cycle that is executing (cycle.rs:49 ran before the opcode), so
    // a frame change makes the next cycle fetch exactly when the page differs (cycle.rs:59)
    const u32 old_page = CF(sh, s, CF_CODE_PAGE), new_page = e[E_CODE_PAGE];
    CF(sh, s, CF_CODE_PAGE) = new_page;
    if (!(s.kflags & KF_CODE_PAGE_CHANGED)) CF(sh, s, CF_PREV_CODE_PAGE) = old_page;
    s.kflags = new_page != CF(sh, s, CF_PREV_CODE_PAGE) ? (s.kflags | KF_CODE_PAGE_CHANGED) : (s.kflags & ~KF_CODE_PAGE_CHANGED);
  }
  const u32 sppc = e[E_SP_PC];
  s.sp = sppc & 0xffffu;
  s.pc = sppc >> 16;
  const u32 ehf = e[E_EH_FLAGS];
  s.kflags &= ~(KF_KERNEL | KF_STATIC | KF_LOCAL);
  if ((ehf >> 16) & 0xffu) s.kflags |= KF_STATIC;
  if ((ehf >> 24) & 0xffu) s.kflags |= KF_LOCAL;
  s.ergs = e[E_ERGS];
  cfv_set_heap_bound(sh, s, e[E_HEAP_BOUND]);
  cfv_set_aux_bound(sh, s, e[E_AUX_BOUND]);
  if (e[E_THIS] < 0x10000u && (e[E_THIS + 1] | e[E_THIS + 2] | e[E_THIS + 3] | e[E_THIS + 4]) == 0) s.kflags |= KF_KERNEL;  // execution_stack.rs:83-87
  if (same_slot) return;
  const u32 new_slot = e[E_SLOT];
  const uint2 bd = P.blob_dir[e[E_CODE_BLOB]];
  CF(sh, s, CF_CODE_OFF) = bd.x;
  CF(sh, s, CF_CODE_LEN) = bd.y;
  cfv_set_slot(sh, s, new_slot);
  if (fresh_slot) {
    CF(sh, s, CF_STACK_HWM) = 0;
    cfv_set_heap_hwm(sh, s, 0);
    CF(sh, s, CF_AUX_HWM) = 0;
  } else {
    const zkw_dev_frame_meta fm = P.frames[(u64)lane_inst(sh, s) * P.F + new_slot];
    CF(sh, s, CF_STACK_HWM) = fm.stack_hwm;
    cfv_set_heap_hwm(sh, s, fm.heap_hwm);
    CF(sh, s, CF_AUX_HWM) = fm.aux_hwm;
  }
}
// load the hot fields of entry `s.depth` into the lane
ZD void frame_load(ZKW_KP P, const Shared& sh, Lane& s, bool same_slot = false) {
  const u32* e = (const u32*)entry_ptr(P, sh, s, s.depth);
  frame_apply(P, sh, s, e, same_slot, false);
}

// ---------------------------------------------------------------------------------------------
// storage — the device-side InMemoryStorage (testing/storage.rs:79-186): open addressing + journal
// ---------------------------------------------------------------------------------------------
ZD u32 storage_hash(u32 shard, const u32 addr[5], const u256& key) {
  u32 h = 0x9e3779b9u * (shard + 1);
#pragma unroll
  for (int i = 0; i < 8; i++) h = (h ^ key.w[i]) * 0x85ebca6bu, h ^= h >> 15;
#pragma unroll
  for (int i = 0; i < 5; i++) h = (h ^ addr[i]) * 0xc2b2ae35u, h ^= h >> 13;
  return h;
}
// returns the entry index of (shard,address,key), inserting an empty (value 0) entry when absent
ZD u32 storage_find(ZKW_KP P, const Shared& sh, Lane& s, u32 shard, const u32 addr[5], const u256& key) {
  const u32 mask = P.storage_slots - 1;
  u32 i = storage_hash(shard, addr, key) & mask;
  zkw_dev_storage_entry* tab = P.storage + (u64)lane_inst(sh, s) * P.storage_slots;
  for (u32 probe = 0; probe < P.storage_slots; probe++, i = (i + 1) & mask) {
    zkw_dev_storage_entry* e = tab + i;
    // key (2 x 16 B) and address + state (2 x 16 B) in four wide loads issued together: a short-circuit compare of
    // 14 separately loaded dwords is 14 dependent memory round trips
    const uint4* e4 = (const uint4*)e;
    const uint4 k0 = e4[0], k1 = e4[1], a0 = e4[4], a1 = e4[5];
    const u32 st = a1.y;
    if (!(st & 0x100u)) {  // free: claim
      atomicOr(P.storage_dirty + (u64)lane_inst(sh, s) * ((P.storage_slots + 31u) >> 5) + (i >> 5), 1u << (i & 31u));  // fire-and-forget: the next reset restores only marked slots
#pragma unroll
      for (int k = 0; k < 8; k++) {
        e->key[k] = key.w[k];
        e->value[k] = 0;
      }
#pragma unroll
      for (int k = 0; k < 5; k++) e->address[k] = addr[k];
      e->shard_state = shard | 0x100u;
      return i;
    }
    const u32 diff = ((st & 0xffu) ^ shard) | (k0.x ^ key.w[0]) | (k0.y ^ key.w[1]) | (k0.z ^ key.w[2]) | (k0.w ^ key.w[3]) | (k1.x ^ key.w[4]) |
                     (k1.y ^ key.w[5]) | (k1.z ^ key.w[6]) | (k1.w ^ key.w[7]) | (a0.x ^ addr[0]) | (a0.y ^ addr[1]) | (a0.z ^ addr[2]) | (a0.w ^ addr[3]) |
                     (a1.x ^ addr[4]);
    const bool same = diff == 0;
    if (same) return i;
  }
  lane_fail(s, ZKW_STATUS_LIMIT);
  return 0;
}
// Storage::execute_partial_query (storage.rs:88-139) + access_storage's read convention (helpers.rs:145-148)
ZD void access_storage(ZKW_KP P, Shared& sh, Lane& s, LogQ& q) {
  const u32 slot = storage_find(P, sh, s, q.shard_id, q.address, q.key);
  if (!lane_ok(s)) return;
  zkw_dev_storage_entry* e = P.storage + (u64)lane_inst(sh, s) * P.storage_slots + slot;
  u256 cur;
#pragma unroll
  for (int k = 0; k < 8; k++) cur.w[k] = e->value[k];
  atomicOr(P.storage_dirty + (u64)lane_inst(sh, s) * ((P.storage_slots + 31u) >> 5) + (slot >> 5), 1u << (slot & 31u));
  e->shard_state |= 0x200u;  // warm marker
  q.read_value = cur;
  if (q.rw) {
    if (CF(sh, s, CF_JOURNAL_LEN) >= P.storage_journal) {
      lane_fail(s, ZKW_STATUS_LIMIT);
      return;
    }
    zkw_dev_journal_entry* j = P.journal + (u64)lane_inst(sh, s) * P.storage_journal + CF(sh, s, CF_JOURNAL_LEN);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      j->old_value[k] = cur.w[k];
      e->value[k] = q.written_value.w[k];
    }
    j->slot = slot;
    CF(sh, s, CF_JOURNAL_LEN)++;
    e->shard_state |= 0x400u;  // the key now exists in the reference's `inner` map and stays there across rollbacks
  } else {
    q.written_value = q.read_value;
  }
  emit_log(P, sh, s, q, ZKW_LQ_LOG);
}
// Storage::finish_frame(panicked) (storage.rs:144-186): undo this frame's writes newest-first
ZD void storage_finish_frame(ZKW_KP P, const Shared& sh, Lane& s, u32 mark, bool panicked) {
  if (!panicked) return;
  while (CF(sh, s, CF_JOURNAL_LEN) > mark) {
    CF(sh, s, CF_JOURNAL_LEN)--;
    const zkw_dev_journal_entry* j = P.journal + (u64)lane_inst(sh, s) * P.storage_journal + CF(sh, s, CF_JOURNAL_LEN);
    zkw_dev_storage_entry* e = P.storage + (u64)lane_inst(sh, s) * P.storage_slots + j->slot;
#pragma unroll
    for (int k = 0; k < 8; k++) e->value[k] = j->old_value[k];
  }
}

// ---------------------------------------------------------------------------------------------
// helpers shared by opcodes
// ---------------------------------------------------------------------------------------------
ZD u32 clip16(const Shared& sh, const u256& v) {  // AllowedPcOrImm::from_u64_clipped(value.low_u64())
  if (sh.clip_mode == 0) return (v.w[1] != 0 || v.w[0] > 0xffffu) ? 0xffffu : v.w[0];
  return v.w[0] & 0xffffu;
}

struct Operand {
  bool has_loc;
  u32 type, page, index;
};

// MemOpsProcessor::compute_addresses_and_select_operands (mem_ops.rs:14-125)
ZD Operand compute_address(ZKW_KP P, const Shared& sh, Lane& s, u32& sp, const u256& reg_value, u32 imm, u32 mode, bool is_write) {
  Operand o;
  o.has_loc = false;
  o.type = ZKW_MEM_STACK;
  o.page = 0;
  o.index = 0;
  // (`mode` is wave-uniform: a register / immediate operand costs neither the clip nor the LDS read of the base page)
  if (ZKW_LIKELY(mode != ZKW_MODE_STACK_PP && mode != ZKW_MODE_STACK_OFF && mode != ZKW_MODE_CODE && mode != ZKW_MODE_STACK_ABS)) return o;
  const u32 vaddr = (clip16(sh, reg_value) + imm) & 0xffffu;  // :34-35
  if (mode == ZKW_MODE_CODE) {  // :100-110
    o.type = ZKW_MEM_CODE;
    o.page = CF(sh, s, CF_CODE_PAGE);
    o.index = vaddr;
    o.has_loc = true;
    return o;
  }
  o.page = CF(sh, s, CF_BASE_PAGE) + 1;  // stack_page_from_base
  if (mode == ZKW_MODE_STACK_PP) {
    if (is_write) {  // :55-70
      o.index = sp;
      sp = (sp + vaddr) & 0xffffu;
    } else {  // :71-86
      sp = (sp - vaddr) & 0xffffu;
      o.index = sp;
    }
    o.has_loc = true;
  } else if (mode == ZKW_MODE_STACK_OFF) {  // :88-98
    o.index = (sp - vaddr) & 0xffffu;
    o.has_loc = true;
  } else {  // ZKW_MODE_STACK_ABS :111-121
    o.index = vaddr;
    o.has_loc = true;
  }
  return o;
}

// perform_dst0_update (helpers.rs:266-283)
template <class RF>
ZD void dst0_update(ZKW_KP P, Shared& sh, RF& rf, Lane& s, const Operand& dst0, u32 dst0_idx, const u256& v, bool is_ptr) {
  s.lane = zkw_lane_id();  // fresh, short-lived lane index (see struct Lane)
  if (ZKW_UNLIKELY(dst0.has_loc)) {
    stack_write(P, sh, s, dst0.index, v, is_ptr);
    emit_mem(P, sh, s, s.timestamp + 3, ZKW_MEM_STACK, dst0.page, dst0.index, v, is_ptr, true, 0);
  } else {
    reg_write(sh, rf, s, dst0_idx, v, is_ptr);
  }
}

ZD void set_flags3(Lane& s, bool lt, bool eq, bool gt) {
  s.flags = (s.flags & FLAG_PENDING) | (lt ? FLAG_LT : 0u) | (eq ? FLAG_EQ : 0u) | (gt ? FLAG_GT : 0u);
}

struct FatPtr {  // zkevm_opcode_defs::FatPointer (Appendix B layout)
  u32 offset, page, start, length;
};
ZD FatPtr fat_ptr_from(const u256& v) {
  FatPtr p;
  p.offset = v.w[0]; p.page = v.w[1]; p.start = v.w[2]; p.length = v.w[3];
  return p;
}
ZD u256 fat_ptr_to_u256(const FatPtr& p) {
  u256 v = u256_zero();
  v.w[0] = p.offset; v.w[1] = p.page; v.w[2] = p.start; v.w[3] = p.length;
  return v;
}
#define FPV_OFFSET_NOT_ZERO 1u
#define FPV_DEREF_BEYOND 2u
ZD u32 fat_ptr_validate(const FatPtr& p, bool fresh) {
  u32 e = 0;
  if (fresh && p.offset != 0) e |= FPV_OFFSET_NOT_ZERO;
  if (p.start + p.length < p.start) e |= FPV_DEREF_BEYOND;
  return e;
}
// FarCallForwardPageType of an ABI byte -> 0 UseHeap, 1 ForwardFatPointer, 2 UseAuxHeap; `codes` = zkw_isa_consts.forwarding_codes
ZD u32 forward_type(u32 codes, u32 b) { return b == ((codes >> 8) & 0xffu) ? 1u : (b == ((codes >> 16) & 0xffu) ? 2u : 0u); }

// build a callstack entry image (32 dwords) from the lane's current frame: cold fields come from HBM
ZD void entry_image_current(ZKW_KP P, const Shared& sh, const Lane& s, u32 img[32]) {
  const uint4* e = entry_ptr(P, sh, s, s.depth);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint4 v = e[i];
    img[4 * i] = v.x; img[4 * i + 1] = v.y; img[4 * i + 2] = v.z; img[4 * i + 3] = v.w;
  }
  img[E_SP_PC] = (s.sp & 0xffffu) | (s.pc << 16);
  img[E_ERGS] = s.ergs;
  img[E_HEAP_BOUND] = cfv_heap_bound(sh, s);
  img[E_AUX_BOUND] = cfv_aux_bound(sh, s);
}

// VmState::start_frame (helpers.rs:225-246): Storage/EventSink::start_frame are a journal mark here
// (the event sink is replayed on the host); emits WT.start_new_execution_context and pushes.
ZD void start_frame(ZKW_KP P, Shared& sh, Lane& s, const u32 prev[32], u32 next[32], bool far) {
  next[E_JOURNAL_MARK] = CF(sh, s, CF_JOURNAL_LEN);
  uint4* a = aux_alloc(P, sh, s, ZKW_AUX_FRAME_START, far ? 1u : 0u, 0, 0, 0);
  if (a) {
#pragma unroll
    for (int i = 0; i < 7; i++) {
      a[1 + i] = make_uint4(prev[4 * i], prev[4 * i + 1], prev[4 * i + 2], prev[4 * i + 3]);
      a[8 + i] = make_uint4(next[4 * i], next[4 * i + 1], next[4 * i + 2], next[4 * i + 3]);
    }
    // (the unused tail of an aux record is not written: zkw_batch_get_instance_trace zeroes it by record type)
  }
  if (s.depth + 1 > P.D) {
    lane_fail(s, ZKW_STATUS_LIMIT);
    return;
  }
  // callstack.push_entry: the old current (with its hot fields) stays at [depth], the new one goes to [depth+1]
  uint4* cur = entry_ptr(P, sh, s, s.depth);
  uint4* nxt = entry_ptr(P, sh, s, s.depth + 1);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    cur[i] = make_uint4(prev[4 * i], prev[4 * i + 1], prev[4 * i + 2], prev[4 * i + 3]);
    nxt[i] = make_uint4(next[4 * i], next[4 * i + 1], next[4 * i + 2], next[4 * i + 3]);
  }
  if (far) hwm_writeback(P, sh, s);  // (a near-call frame keeps the slot and its marks)
  s.depth++;
  frame_apply(P, sh, s, next, !far, far);
}

// =============================================================================================
// opcodes
// =============================================================================================

struct Decoded {
  u32 word_lo, word_hi;  // the 64-bit instruction word
  u32 attr;
  u32 cond, src0, src1, dst0, dst1, imm0, imm1;
};
struct Pre {  // PreState (cycle.rs:8-14)
  u256 src0, src1;
  // (0 / 1 in vector registers, not `bool`: as lane masks carried through the divergent operand phase of the variant-group
  // instantiation they were mis-merged — an operand that was no pointer lost its fat-pointer metadata)
  u32 src0_ptr, src1_ptr;
  Operand dst0;
  u32 new_pc;
};

// The heavy opcode bodies (near_call, log, far_call, ret) run out of line (zkw_heavy_entry) and have no access to the
// VGPR register file: what they want written to registers comes back as a HeavyOut and is applied by the caller.
#define ZKW_ACT_DST0 1u       /* perform_dst0_update(v1) */
#define ZKW_ACT_FAR 2u        /* far_call.rs:573-610: r1 = v1 (pointer), r2 = v2, r3..r12 cleared unless ZKW_ACT_TO_SYSTEM, r13..r15 = 0 */
#define ZKW_ACT_TO_SYSTEM 4u
#define ZKW_ACT_RET 8u        /* ret.rs:213-233: r1 = v1 (pointer), r2..r15 = 0 */
struct HeavyOut {
  u256 v1, v2;
  u32 action;
};

// near_call.rs:6-68
ZD void op_near_call(ZKW_KP P, Shared& sh, Lane& s, const Decoded& d, const Pre& ps) {
  s.flags &= FLAG_PENDING;  // reset_flags
  const u32 abi_ergs = ps.src0.w[0];
  const u32 remaining = s.ergs;
  u32 passed, left;
  if (abi_ergs == 0 || remaining < abi_ergs) {
    passed = remaining;
    left = 0;
  } else {
    passed = abi_ergs;
    left = remaining - abi_ergs;
  }
  s.ergs = left;
  s.pc = ps.new_pc;
  u32 prev[32], next[32];
  entry_image_current(P, sh, s, prev);
#pragma unroll
  for (int i = 0; i < 32; i++) next[i] = prev[i];
  next[E_SP_PC] = (s.sp & 0xffffu) | (d.imm0 << 16);
  next[E_EH_FLAGS] = (d.imm1 & 0xffffu) | (prev[E_EH_FLAGS] & 0x00ff0000u) | (1u << 24);  // is_local_frame = true
  next[E_ERGS] = passed;
  start_frame(P, sh, s, prev, next, false);
}

// context.rs:6-111
template <class RF>
ZD void op_context(ZKW_KP P, Shared& sh, RF& rf, Lane& s, const Decoded& d, const Pre& ps) {
  s.pc = ps.new_pc;
  const u32 v = ZKW_ATTR_VARIANT(d.attr);
  u256 value = u256_zero();
  if (v == ZKW_CTX_SET_CONTEXT_U128) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      changed = changed || CF(sh, s, CF_CTX0 + i) != ps.src0.w[i];
      CF(sh, s, CF_CTX0 + i) = ps.src0.w[i];
    }
    if (changed) s.kflags |= KF_COLD_DIRTY;
    return;
  }
  if (v == ZKW_CTX_SET_ERGS_PER_PUBDATA) {
    if (CF(sh, s, CF_ERGS_PP) != ps.src0.w[0]) s.kflags |= KF_COLD_DIRTY;
    CF(sh, s, CF_ERGS_PP) = ps.src0.w[0];
    return;
  }
  if (v == ZKW_CTX_INC_TX_NUMBER) {
    CF(sh, s, CF_TX_NUMBER) = (CF(sh, s, CF_TX_NUMBER) + 1) & 0xffffu;
    s.kflags |= KF_COLD_DIRTY;
    return;
  }
  const u32* e = (const u32*)entry_ptr(P, sh, s, s.depth);
  if (v == ZKW_CTX_THIS || v == ZKW_CTX_CALLER || v == ZKW_CTX_CODE_ADDRESS) {
    const u32 off = v == ZKW_CTX_THIS ? E_THIS : (v == ZKW_CTX_CALLER ? E_SENDER : E_CODE_ADDR);
#pragma unroll
    for (int i = 0; i < 5; i++) value.w[i] = e[off + i];
  } else if (v == ZKW_CTX_META) {  // VmMetaParameters::to_u256 (Appendix B layout)
    const u32 sh3 = e[E_SHARDS];
    value.w[0] = CF(sh, s, CF_ERGS_PP);
    value.w[2] = cfv_heap_bound(sh, s);
    value.w[3] = cfv_aux_bound(sh, s);
    value.w[7] = (sh3 & 0xffu) | (((sh3 >> 8) & 0xffu) << 8) | (((sh3 >> 16) & 0xffu) << 16);
  } else if (v == ZKW_CTX_ERGS_LEFT) {
    value.w[0] = s.ergs;
  } else if (v == ZKW_CTX_SP) {
    value.w[0] = s.sp;
  } else {  // GetContextU128
#pragma unroll
    for (int i = 0; i < 4; i++) value.w[i] = e[E_CTX + i];
  }
  ZKW_SETTLE(3 /* context */);
  dst0_update(P, sh, rf, s, ps.dst0, d.dst0, value, false);
}

// ptr.rs:6-194
template <class RF>
ZD void op_ptr(ZKW_KP P, Shared& sh, RF& rf, Lane& s, const Decoded& d, const Pre& ps) {
  ZKW_DIV_SCOPE;  // (lanes return early with a pending exception; the others may emit a stack write below)
  s.pc = ps.new_pc;
  const u32 v = ZKW_ATTR_VARIANT(d.attr);
  if (!ps.src0_ptr || ps.src1_ptr) {  // :35-45
    s.flags |= FLAG_PENDING;
    return;
  }
  u256 result = ps.src0;
  if (v == ZKW_PTR_ADD || v == ZKW_PTR_SUB) {
    const u64 max_off = P.consts.max_offset_for_add_sub;  // ptr::MAX_OFFSET_FOR_ADD_SUB (:47)
    const bool too_far = (ps.src1.w[2] | ps.src1.w[3] | ps.src1.w[4] | ps.src1.w[5] | ps.src1.w[6] | ps.src1.w[7]) != 0 || (((u64)ps.src1.w[1] << 32) | ps.src1.w[0]) >= max_off;
    const u32 off = ps.src1.w[0];
    const u32 cur = ps.src0.w[0];
    u32 n;
    bool err;
    if (v == ZKW_PTR_ADD) {
      n = cur + off;
      err = n < cur;
    } else {
      n = cur - off;
      err = cur < off;
    }
    if (too_far || err) {
      s.flags |= FLAG_PENDING;
      return;
    }
    result.w[0] = n;  // :82 low 128 bits from the pointer, high 128 from src0
  } else if (v == ZKW_PTR_PACK) {
    if ((ps.src1.w[0] | ps.src1.w[1] | ps.src1.w[2] | ps.src1.w[3]) != 0) {  // :110-114
      s.flags |= FLAG_PENDING;
      return;
    }
#pragma unroll
    for (int i = 4; i < 8; i++) result.w[i] = ps.src1.w[i];  // :126
  } else {  // Shrink
    const u32 off = ps.src1.w[0];
    if (ps.src0.w[3] < off) {
      s.flags |= FLAG_PENDING;
      return;
    }
    result.w[3] = ps.src0.w[3] - off;
  }
  dst0_update(P, sh, rf, s, ps.dst0, d.dst0, result, true);
}

#ifdef __HIP_DEVICE_COMPILE__
// Heap prefetch for the instructions behind the current one.  A heap word comes from HBM (the arenas of a launch are
// hundreds of MB): ~2000 clocks between the issue of a UMA's word loads and their return, a wave per SIMD and nothing to
// overlap them with — the largest single wait of a UMA cycle.  When a code word is fetched its four
// opcodes are decoded at once, so the instructions behind the current one are known up to three cycles ahead, and when
// one of them is a heap access with an immediate address (src0 = UseImm16Only: the offset is in the instruction, no
// register involved) its words can be requested now: two `global_load_lds_dword` per word (one per 16-byte plane: each
// lane touches its own 64-byte segment), results written to a 256-byte sink in LDS that nobody reads — a load without a
// destination register, so nothing to keep alive and nothing to wait for: the lines are in L2 when the access itself
// comes.  A hint only: a jump in between, a frame change or a word beyond the written part of the heap make it useless
// or skip it, never wrong.  (Not for register-addressed accesses — the register may be written by an instruction in
// between — nor across code words: op_uma requests those itself, early in their own cycle.)  Same-box A/B on the
// driver's command: +3 % (profiles/r05_prefetch_ab.txt); checking the next opcode every cycle instead (an LDS read of its
// slot) gave the same.
// the words an access at byte offset `off` of a heap / aux-heap page touches (page in arena slot `slot`, written up to `hwm`)
ZD void prefetch_page_words(const uint4* arena, u32 words_per_page, u32 L, u32 lds_sink, u32 slot, u32 hwm, u32 off) {
  const u32 lane = zkw_lane_id();
  const u32 w0 = off >> 5;
  const u32 last = (off & 31u) ? w0 + 1u : w0;
  for (u32 w = w0; w <= last; w++) {
    if (w >= hwm) continue;  // words at and beyond the mark (<= the page size) read as zero without a memory access
    // (slot * words + w) * L with the scalars spelled out as scalar operands: left to itself the optimiser hoists vector
    // copies of them out of the cycle loop and reloads them from scratch memory here — behind an s_waitcnt vmcnt(0), i.e.
    // behind the stores of the previous cycle
    u32 row, first;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(slot), "s"(words_per_page), "v"(w));
    asm("v_mul_lo_u32 %0, %1, %2" : "=v"(first) : "v"(row), "s"(L));
    const uint4* p = arena + ((u64)2u * first + lane);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\tglobal_load_lds_dword %1, off" : : "v"(p), "v"(p + L), "s"(lds_sink) : "m0");
  }
}
// LDS address of the sink: right behind the ISA table in the dynamic segment (zkw_launch_cycle_kernel: args.lds_sink) — a
// link-time constant, so neither a register nor an LDS word has to carry it to the opcode bodies
ZD u32 zkw_lds_sink_addr() { return (u32)(size_t)(ZKW_LDS_AS char*)((char*)zkw_lds + ZKW_ISA_TABLE_SIZE * 8u); }
ZD void prefetch_uma_words(ZKW_KP P, const Shared& sh, Lane& s, u32 lds_sink, u32 attr, u32 word_hi) {
  const u32 v = ZKW_ATTR_VARIANT(attr);
  if (ZKW_ATTR_OPCODE(attr) != ZKW_OP_UMA || ZKW_ATTR_SRC0(attr) != ZKW_MODE_IMM || (v != ZKW_UMA_HEAP_READ && v != ZKW_UMA_HEAP_WRITE)) return;
  prefetch_page_words(sh.heap, P.H, P.L, lds_sink, cfv_slot(sh, s), cfv_heap_hwm(sh, s), word_hi & 0xffffu);
}
#endif

// uma.rs:26-425
template <class RF>
ZD void op_uma(ZKW_KP P, Shared& sh, RF& rf, Lane& s, const Decoded& d, const Pre& ps) {
  ZKW_SUB_DECL
  s.lane = zkw_lane_id();
  const u32 v = ZKW_ATTR_VARIANT(d.attr);
  s.pc = ps.new_pc;
  const bool increment = ZKW_ATTR_FLAGS(d.attr) & 1u;
  FatPtr fp = fat_ptr_from(ps.src0);
  u32 exceptions = 0;
  bool skip_legit = false;
  const bool is_ptr_read = v == ZKW_UMA_FAT_PTR_READ;
  const bool is_heap = v == ZKW_UMA_HEAP_READ || v == ZKW_UMA_HEAP_WRITE;
  const bool is_write = v == ZKW_UMA_HEAP_WRITE || v == ZKW_UMA_AUX_WRITE;
  if (is_ptr_read && !ps.src0_ptr) exceptions |= 1u;  // INPUT_IS_NOT_POINTER_WHEN_EXPECTED :73-78
  // the frame fields this access needs, read from LDS together (one wait instead of one per use; the variant is
  // wave-uniform, so these selections are scalar branches)
  u32 f_base_page = 0, f_slot = 0, f_hwm = 0, f_bound = 0;
  if (ZKW_LIKELY(!is_ptr_read)) {
    f_base_page = CF(sh, s, CF_BASE_PAGE);
    f_slot = cfv_slot(sh, s);
    f_hwm = is_heap ? cfv_heap_hwm(sh, s) : CF(sh, s, CF_AUX_HWM);
    f_bound = is_heap ? cfv_heap_bound(sh, s) : cfv_aux_bound(sh, s);
    ZKW_LGKM_PROBE(7 /* UMA frame fields */)
  }
  const u32 f_hwm_in = f_hwm;
  // The word loads of a heap / aux-heap access are ISSUED here, in front of the exception / growth / cost arithmetic (~900
  // clocks that need nothing from memory), and waited for behind it: the address needs only the offset and the frame fields
  // above.  Speculative for the lanes that turn out to skip the access (an exception): those either hold an offset beyond
  // 2^32 (no load: the guard below) or read a word of their own page that nobody looks at.
  // (straight into the words the access works on: a lane that turns out to skip a heap / aux access has set_panic and
  // nobody looks at what it loaded; a fat-pointer read loads behind the checks, below)
  u256 w0v = u256_zero(), w1v = u256_zero();
  if (ZKW_LIKELY(!is_ptr_read)) {
    if (ZKW_LIKELY(!(ps.src0.w[1] | ps.src0.w[2] | ps.src0.w[3] | ps.src0.w[4] | ps.src0.w[5] | ps.src0.w[6] | ps.src0.w[7]))) {
      const u32 ew = ps.src0.w[0] >> 5;
      w0v = heap_read_at(P, sh, s, !is_heap, f_slot, f_hwm, ew);
      if (ps.src0.w[0] & 31u) w1v = heap_read_at(P, sh, s, !is_heap, f_slot, f_hwm, ew + 1u);
    }
  }
  u32 mem_type;
  if (is_ptr_read) {
    mem_type = ZKW_MEM_FAT_PTR;
  } else if (is_heap) {
    fp.page = f_base_page + 2;
    mem_type = ZKW_MEM_HEAP;
  } else {
    fp.page = f_base_page + 3;
    mem_type = ZKW_MEM_AUX_HEAP;
  }
  u32 src_offset;
  if (is_ptr_read) {  // :110-120
    if (!(fp.offset < fp.length)) skip_legit = true;
    src_offset = fp.start + fp.offset;
  } else {  // :121-135  src0 > MAX_OFFSET_TO_DEREF
    const bool beyond = (ps.src0.w[1] | ps.src0.w[2] | ps.src0.w[3] | ps.src0.w[4] | ps.src0.w[5] | ps.src0.w[6] | ps.src0.w[7]) != 0 ||
                        ps.src0.w[0] > sh.max_deref_low;
    if (ZKW_UNLIKELY(beyond)) {
      exceptions |= 2u;  // DEREF_BEYOND_HEAP_RANGE
      skip_legit = true;
    }
    src_offset = fp.offset;
  }
  const u32 incremented = fp.offset + 32u;
  if (ZKW_UNLIKELY(incremented < fp.offset)) {  // :139-147
    exceptions |= 4u;
    if (!is_ptr_read && !(exceptions & 2u)) lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
  }
  u32 growth = 0;  // :152-194
  if (!is_ptr_read) {
    const u32 bound = f_bound;
    if (ZKW_UNLIKELY(incremented >= bound)) {
      growth = incremented - bound;
      if (is_heap) cfv_set_heap_bound(sh, s, incremented); else cfv_set_aux_bound(sh, s, incremented);
      s.kflags |= KF_TAIL2;
    }
  }
  u32 cost = growth * sh.growth_per_byte;  // :196-197
  if (exceptions & 2u) cost = 0xffffffffu;                   // :202-207
  if (ZKW_UNLIKELY(s.ergs < cost)) {
    s.ergs = 0;
    exceptions |= 8u;  // NOT_ENOUGH_ERGS_TO_GROW_MEMORY
  } else {
    s.ergs -= cost;
  }
  const bool set_panic = exceptions != 0;
  const bool skip = skip_legit || set_panic;  // :223-228
  const u32 word0 = src_offset >> 5, word1 = word0 + 1, unal = src_offset & 31u;
  const bool unaligned = unal != 0;
  const u32 ts_r = s.timestamp, ts_w = s.timestamp + 3;
  ZKW_SUB(41)  // exceptions, growth, cost
#ifdef ZKW_PROFILE_DRAIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ZKW_SUB(70)  // drain of the stores issued before this point
#endif
  ZKW_DIV_IF(ZKW_LIKELY(!skip)) {  // :265-288
    // both word loads are issued before the first query is emitted: the emission needs the loaded value, so reading
    // and emitting word by word would serialise two memory round trips (the dominant cost of this opcode)
    if (is_ptr_read) {
      w0v = fat_ptr_read(P, sh, s, fp.page, word0);
      if (unaligned) w1v = fat_ptr_read(P, sh, s, fp.page, word1);
    }  // (else: requested above — an access that is not skipped has an offset below 2^32: these are its words)
    ZKW_SETTLE(2 /* UMA words */);
    ZKW_SUB(64)  // loads issued
#ifdef ZKW_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ZKW_SUB(65)  // loads returned
#endif
    emit_mem(P, sh, s, ts_r, mem_type, fp.page, word0, w0v, false, false, 0);
    ZKW_SUB(66)  // first read query
    ZKW_DIV_IF(unaligned) emit_mem(P, sh, s, ts_r, mem_type, fp.page, word1, w1v, false, false, 0);
  }
  ZKW_SUB(42)  // word reads + read queries
  // A wave whose lanes all access at a word boundary (a shared tape with an immediate or a common cursor: half of cfg 2's
  // accesses) needs neither the byte window of a read (~50 instructions) nor the six 256-bit shifts that merge a written
  // value into two words (~270): the word is the value.  One ballot decides; the general code gives the same for unal = 0.
  const bool all_aligned = zkw_ballot(unal != 0) == 0;
  // ... and a wave whose lanes share one unaligned offset shifts by scalar amounts with static indices (u256_byte_window_at /
  // u256_merge_at): the dword part of the offset selects the case, a scalar branch
  const u32 u_unal = (u32)__builtin_amdgcn_readfirstlane((int)unal);
  const bool same_unal = zkw_ballot(unal != u_unal) == 0;
  const u32 u_b8 = (u_unal & 3u) * 8u;
  if (!is_write) {  // :291-348
    u256 result;
    if (all_aligned) {
      result = w0v;
    } else if (same_unal) {
      switch (u_unal >> 2) {
        case 0: result = u256_byte_window_at<0>(w0v, w1v, u_b8); break;
        case 1: result = u256_byte_window_at<1>(w0v, w1v, u_b8); break;
        case 2: result = u256_byte_window_at<2>(w0v, w1v, u_b8); break;
        case 3: result = u256_byte_window_at<3>(w0v, w1v, u_b8); break;
        case 4: result = u256_byte_window_at<4>(w0v, w1v, u_b8); break;
        case 5: result = u256_byte_window_at<5>(w0v, w1v, u_b8); break;
        case 6: result = u256_byte_window_at<6>(w0v, w1v, u_b8); break;
        default: result = u256_byte_window_at<7>(w0v, w1v, u_b8); break;
      }
    } else {
      result = u256_byte_window(w0v, w1v, unal);
    }
    if (is_ptr_read) {
      u32 beyond = incremented - fp.length;
      if (incremented < fp.length || skip) beyond = 0;
      beyond &= 31u;
      result = u256_select_bits(u256_low_mask(beyond * 8), u256_zero(), result);  // the low `beyond` bytes read as zero
    }
    ZKW_SUB(43)  // read: shifts
    ZKW_DIV_IF(ZKW_LIKELY(!set_panic)) {
      dst0_update(P, sh, rf, s, ps.dst0, d.dst0, result, false);
      if (increment) {
        u256 upd = ps.src0;
        upd.w[0] = incremented;  // (l[0] & TOP_32) + incremented :337-338
        reg_write(sh, rf, s, d.dst1, upd, ps.src0_ptr);
      }
    } else {
      s.flags |= FLAG_PENDING;
    }
    ZKW_SUB(44)  // read: destination
  } else {  // :349-423
    u256 n0, n1;
    if (all_aligned) {
      n0 = ps.src1;
      n1 = u256_zero();
    } else if (same_unal) {
      switch (u_unal >> 2) {
        case 0: u256_merge_at<0>(w0v, w1v, ps.src1, u_b8, n0, n1); break;
        case 1: u256_merge_at<1>(w0v, w1v, ps.src1, u_b8, n0, n1); break;
        case 2: u256_merge_at<2>(w0v, w1v, ps.src1, u_b8, n0, n1); break;
        case 3: u256_merge_at<3>(w0v, w1v, ps.src1, u_b8, n0, n1); break;
        case 4: u256_merge_at<4>(w0v, w1v, ps.src1, u_b8, n0, n1); break;
        case 5: u256_merge_at<5>(w0v, w1v, ps.src1, u_b8, n0, n1); break;
        case 6: u256_merge_at<6>(w0v, w1v, ps.src1, u_b8, n0, n1); break;
        default: u256_merge_at<7>(w0v, w1v, ps.src1, u_b8, n0, n1); break;
      }
    } else {
      const u32 lowest = 32 - unal;
      n0 = u256_shl(u256_shr(w0v, lowest * 8), lowest * 8);
      n0 = u256_or(n0, u256_shr(ps.src1, unal * 8));
      n1 = u256_shr(u256_shl(w1v, unal * 8), unal * 8);
      n1 = u256_or(n1, u256_shl(ps.src1, (32 - unal) * 8));
    }
    ZKW_SUB(45)  // write: shifts
    ZKW_DIV_IF(ZKW_LIKELY(!skip)) {
      heap_write_at(P, sh, s, !is_heap, f_slot, f_hwm, word0, n0);
      emit_mem(P, sh, s, ts_w, mem_type, fp.page, word0, n0, false, true, 0);
      ZKW_DIV_IF(unaligned) {
        heap_write_at(P, sh, s, !is_heap, f_slot, f_hwm, word1, n1);
        emit_mem(P, sh, s, ts_w, mem_type, fp.page, word1, n1, false, true, 0);
      }
    }
    if (ZKW_UNLIKELY(f_hwm != f_hwm_in)) {
      if (is_heap) cfv_set_heap_hwm(sh, s, f_hwm); else CF(sh, s, CF_AUX_HWM) = f_hwm;
    }
    ZKW_SUB(46)  // write: heap writes + write queries
    ZKW_DIV_IF(ZKW_LIKELY(!set_panic)) {
      if (increment) {
        u256 upd = ps.src0;
        upd.w[0] = incremented;
        dst0_update(P, sh, rf, s, ps.dst0, d.dst0, upd, false);
      }
    } else {
      s.flags |= FLAG_PENDING;
    }
    ZKW_SUB(47)  // write: destination
  }
#ifdef __HIP_DEVICE_COMPILE__
  // A heap access through a register with post-increment is a cursor: the next access through it is 32 bytes on, and its
  // words can be requested now — cycles before that instruction arrives (the code-word prefetch cannot see a register-held
  // address).  A hint like the others: wrong guesses cost two loads into the sink.
  if (!is_ptr_read && increment && !set_panic && ZKW_ATTR_SRC0(d.attr) != ZKW_MODE_IMM && !ZKW_ABL(sh.debug_flags, ZKW_NO_PREFETCH))
    prefetch_page_words(is_heap ? sh.heap : sh.aux_heap, is_heap ? P.H : P.A, P.L, zkw_lds_sink_addr(), f_slot, f_hwm, incremented);
#endif
}

// log.rs:11-330 (precompile calls: see zkw_precompiles below)
ZD void call_precompile(ZKW_KP P, Shared& sh, Lane& s, const LogQ& q);

// KIND: 0 = storage read / write, 1 = event / L1 message, 2 = precompile call — one out-of-line function each
// (zkw_heavy_log*), so that the storage probe and the 40-dword query do not share a register budget and a set of
// callee-saved registers with the precompile call; the variant checks below then fold away
template <int KIND>
ZD void op_log(ZKW_KP P, Shared& sh, Lane& s, const Decoded& d, const Pre& ps, HeavyOut& out) {
  ZKW_DIV_SCOPE;  // (lanes return early — not enough ergs, a failed storage probe — while the others emit their query)
  const u32 v = ZKW_ATTR_VARIANT(d.attr);
  if (KIND == 0 && !(v == ZKW_LOG_STORAGE_READ || v == ZKW_LOG_STORAGE_WRITE)) return;
  if (KIND == 1 && !(v == ZKW_LOG_EVENT || v == ZKW_LOG_TO_L1)) return;
  if (KIND == 2 && (v == ZKW_LOG_STORAGE_READ || v == ZKW_LOG_STORAGE_WRITE || v == ZKW_LOG_EVENT || v == ZKW_LOG_TO_L1)) return;
  s.pc = ps.new_pc;
  const bool is_first = ZKW_ATTR_FLAGS(d.attr) & 1u;
  const u32* e = (const u32*)entry_ptr(P, sh, s, s.depth);
  const u32 shard = e[E_SHARDS] & 0xffu;
  const u32 ergs_available = s.ergs;
  const zkw_isa_consts ZKW_CONST_AS& K = P.consts;
  LogQ q;
  q.timestamp = s.timestamp + 1;
  q.tx_number = CF(sh, s, CF_TX_NUMBER);
  q.shard_id = shard;
#pragma unroll
  for (int i = 0; i < 5; i++) q.address[i] = e[E_THIS + i];
  q.key = ps.src0;
  q.read_value = u256_zero();
  q.written_value = ps.src1;
  q.rollback = false;
  u32 ergs_on_pubdata = 0;
  if (v == ZKW_LOG_STORAGE_WRITE) {  // :71-118
    q.aux_byte = K.storage_aux_byte;
    q.rw = true;
    q.is_service = false;
    emit_log(P, sh, s, q, ZKW_LQ_REFUND);  // refund_for_partial_query: InMemoryStorage refunds nothing (storage.rs:80-86)
    const u32 net = shard == 0 ? K.initial_storage_write_pubdata_bytes : 0u;
    ergs_on_pubdata = CF(sh, s, CF_ERGS_PP) * net;
  } else if (v == ZKW_LOG_TO_L1) {
    ergs_on_pubdata = CF(sh, s, CF_ERGS_PP) * K.l1_message_pubdata_bytes;
  }
  const u32 extra = v == ZKW_LOG_PRECOMPILE ? ps.src1.w[0] : 0u;
  const u32 total = extra + ergs_on_pubdata;
  const bool not_enough = ergs_available < total;
  u32 spent;
  if (not_enough) {  // :136-153
    s.ergs = 0;
    spent = ergs_available < ergs_on_pubdata ? ergs_available : ergs_on_pubdata;
  } else {
    s.ergs = ergs_available - total;
    spent = ergs_on_pubdata;
  }
  if (spent) {
    CF(sh, s, CF_SPENT_PUBDATA) += spent;
    s.kflags |= KF_COLD_DIRTY;
  }
  q.is_service = is_first;
  if (v == ZKW_LOG_STORAGE_READ) {  // :163-195
    if (not_enough) {
      lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
      return;
    }
    q.aux_byte = K.storage_aux_byte;
    q.rw = false;
    q.written_value = u256_zero();
    access_storage(P, sh, s, q);
    out.v1 = q.read_value;
    out.action = ZKW_ACT_DST0;
  } else if (v == ZKW_LOG_STORAGE_WRITE) {  // :196-220
    if (not_enough) return;
    access_storage(P, sh, s, q);
  } else if (v == ZKW_LOG_EVENT || v == ZKW_LOG_TO_L1) {  // :221-251; EventSink is replayed on the host
    if (not_enough) {
      if (v != ZKW_LOG_TO_L1) lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
      return;
    }
    q.aux_byte = v == ZKW_LOG_EVENT ? K.event_aux_byte : K.l1_message_aux_byte;
    q.rw = true;
    emit_log(P, sh, s, q, ZKW_LQ_LOG);
  } else {  // PrecompileCall :252-328
    if (not_enough) {
      out.v1 = u256_zero();
      out.action = ZKW_ACT_DST0;
      return;
    }
    const u32 heap_page = CF(sh, s, CF_BASE_PAGE) + 2;
    if (q.key.w[4] == 0) q.key.w[4] = heap_page;  // memory_page_to_read  :273-283
    if (q.key.w[5] == 0) q.key.w[5] = heap_page;  // memory_page_to_write :285-295
    q.aux_byte = K.precompile_aux_byte;
    q.rw = false;
    q.written_value = u256_zero();
    call_precompile(P, sh, s, q);
    out.v1 = u256_from_u32(1);
    out.action = ZKW_ACT_DST0;
  }
}

// VersionedHashGeneric<ContractCodeSha256> (Appendix B): big-endian byte 0 = version, 1 = marker, 2-3 = words
ZD void versioned_hash(const u256& h, bool& ok, u32& marker, u32& len_words, u256& stored) {
  ok = (h.w[7] >> 24) == 1u;
  marker = (h.w[7] >> 16) & 0xffu;
  len_words = h.w[7] & 0xffffu;
  stored = h;
  stored.w[7] &= 0xff00ffffu;
}

// takes back the decommit this cycle chained into the running commitment (the cycle failed behind it: see op_far_call)
#ifdef ZKW_WIDE
// LDS byte address of this wave's hand-over area to the helper wave (0: no helper in this launch); kept in the first
// dword of the wave's LDS header, which the device build does not use otherwise (the stream cursors live in v128)
#ifdef __HIP_DEVICE_COMPILE__
ZD u32 dq_helper_area(const Shared& sh) { return *ZKW_LDS_WORD((ZKW_LDS_AS u32*)sh.cursor); }
ZD u32 kh_box(const Shared& sh) { return *ZKW_LDS_WORD((ZKW_LDS_AS u32*)sh.cursor + 1); }
#else
ZD u32 dq_helper_area(const Shared& sh) { return *ZKW_LDS_WORD(sh.cursor); }
ZD u32 kh_box(const Shared& sh) { return *ZKW_LDS_WORD(sh.cursor + 1); }
#endif
#endif
ZD void dq_undo(ZKW_KP P, const Shared& sh, Lane& s) {
  if (!(s.kflags & KF_DQ_CHAINED)) return;
  if (sh.debug_flags & ZKW_DQ_HELPER) return;  // handed to the helper wave only when the cycle completes: nothing to take back
  s.kflags &= ~KF_DQ_CHAINED;
  const u32 inst = lane_inst(sh, s);
  u64* tail_p = P.commit_out + ((u64)inst * ZKW_QUEUE_COUNT + ZKW_QUEUE_DECOMMIT) * 4;
  const u64* prev_p = P.dq_prev + (u64)inst * 4;
  tail_p[0] = prev_p[0]; tail_p[1] = prev_p[1]; tail_p[2] = prev_p[2]; tail_p[3] = prev_p[3];
  P.dq_count[inst] -= 1;
}

// far_call.rs:35-613
ZD void op_far_call(ZKW_KP P, Shared& sh, Lane& s, const Decoded& d, const Pre& ps, const u256& r15, HeavyOut& out) {
  ZKW_DIV_SCOPE;  // (lanes return early on a failed probe / an unknown code hash / no arena slot left)
  const zkw_isa_consts ZKW_CONST_AS& K = P.consts;
  const u32 variant = ZKW_ATTR_VARIANT(d.attr);
  s.flags &= FLAG_PENDING;  // :69
  const bool is_static_call = ZKW_ATTR_FLAGS(d.attr) & 1u;
  const bool is_call_shard = ZKW_ATTR_FLAGS(d.attr) & 2u;
  const u32 handler = d.imm0;
  u32 called[5];
#pragma unroll
  for (int i = 0; i < 5; i++) called[i] = ps.src1.w[i];
  const bool dst_is_kernel = called[0] < 0x10000u && (called[1] | called[2] | called[3] | called[4]) == 0;
  FatPtr abi = fat_ptr_from(ps.src0);
  const u32 abi_ergs = ps.src0.w[6];
  const u32 fwd = forward_type(K.forwarding_codes, ps.src0.w[7] & 0xffu);
  const u32 abi_shard = (ps.src0.w[7] >> 8) & 0xffu;
  bool constructor_call = ((ps.src0.w[7] >> 16) & 0xffu) != 0;
  bool to_system = ((ps.src0.w[7] >> 24) & 0xffu) != 0;
  constructor_call = constructor_call && (s.kflags & KF_KERNEL) != 0;  // :85
  to_system = to_system && dst_is_kernel;               // :86
  u32 prev[32];
  entry_image_current(P, sh, s, prev);
  const u32 caller_shard = prev[E_SHARDS] & 0xffu;
  const u32 remaining_ergs = s.ergs;
  const u32 new_code_shard = is_call_shard ? abi_shard : caller_shard;
  const u32 new_this_shard = variant == ZKW_FAR_DELEGATE ? caller_shard : new_code_shard;
  const u32 new_base = CF(sh, s, CF_MPC);  // :118
  u256 code_hash;
  bool map_to_trivial;
  ZKW_DIV_IF(new_code_shard != 0 && !P.props.zkporter_is_available) {  // :123-129
    code_hash = u256_zero();
    map_to_trivial = true;
  } else {
    LogQ q;
    q.timestamp = s.timestamp + 1;
    q.tx_number = CF(sh, s, CF_TX_NUMBER);
    q.aux_byte = K.storage_aux_byte;
    q.shard_id = new_code_shard;
    q.address[0] = K.deployer_address_low;
    q.address[1] = q.address[2] = q.address[3] = q.address[4] = 0;
    q.key = u256_zero();
#pragma unroll
    for (int i = 0; i < 5; i++) q.key.w[i] = called[i];
    q.read_value = u256_zero();
    q.written_value = u256_zero();
    q.rw = false;
    q.rollback = false;
    q.is_service = false;
    access_storage(P, sh, s, q);  // :144-145
    if (!lane_ok(s)) return;
    const bool mask_aa = u256_is_zero(q.read_value) && !dst_is_kernel;
    if (mask_aa) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        code_hash.w[2 * i] = (u32)P.props.default_aa_code_hash.l[i];
        code_hash.w[2 * i + 1] = (u32)(P.props.default_aa_code_hash.l[i] >> 32);
      }
    } else {
      code_hash = q.read_value;
    }
    map_to_trivial = false;
  }
  ZKW_STAMP(57)  // far call: entry image, code-hash storage read
  const u32 candidate_page = map_to_trivial ? 0u : new_base;  // :161-165
  u32 exceptions = 0;
  u32 code_len_words = 0;
  bool ok;
  u32 marker, vlen;
  u256 stored;
  versioned_hash(code_hash, ok, marker, vlen, stored);
  if (ok) {
    const bool at_rest = marker == 0, constructed = marker == 1;
    if (!(at_rest || constructed)) {
      exceptions |= 2u;  // INVALID_CODE_HASH_FORMAT
      code_hash = u256_zero();
    } else if ((!constructor_call && at_rest) || (constructor_call && constructed)) {
      code_hash = stored;
      code_len_words = vlen;
    } else if (!dst_is_kernel) {  // :215-237 degrade to default AA
      u256 aa;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        aa.w[2 * i] = (u32)P.props.default_aa_code_hash.l[i];
        aa.w[2 * i + 1] = (u32)(P.props.default_aa_code_hash.l[i] >> 32);
      }
      bool aok;
      u32 am, al;
      u256 as;
      versioned_hash(aa, aok, am, al, as);
      if (!aok || am != 0) {
        lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
        return;
      }
      code_hash = aa;
      code_len_words = al;
    } else {
      exceptions |= 32u;  // CALL_IN_NOW_CONSTRUCTED_SYSTEM_CONTRACT
      code_hash = u256_zero();
    }
  } else {
    exceptions |= 2u;
    code_hash = u256_zero();
  }
  if (fwd == 1u && !ps.src0_ptr) exceptions |= 1u;  // :255-262
  const u32 pve = fat_ptr_validate(abi, fwd != 1u);
  if (pve) exceptions |= 16u;                             // MALFORMED_ABI_QUASI_POINTER
  if (!(abi.offset <= abi.length)) exceptions |= 16u;     // validate_as_slice :280-282
  if (fwd == 1u) {  // :285-314
    abi.start = abi.start + abi.offset;
    abi.length = abi.length - abi.offset;
    abi.offset = 0;
  } else if (fwd == 0u) {
    abi.page = CF(sh, s, CF_BASE_PAGE) + 2;
  } else {
    abi.page = CF(sh, s, CF_BASE_PAGE) + 3;
  }
  if (exceptions) {  // :321-325
    abi.offset = abi.page = abi.start = abi.length = 0;
  }
  u32 growth = 0;  // :330-369
  if (fwd != 1u) {
    u32 upper = abi.start + abi.length;
    if (pve & FPV_DEREF_BEYOND) upper = 0xffffffffu;
    const u32 bound = fwd == 0u ? cfv_heap_bound(sh, s) : cfv_aux_bound(sh, s);
    if (upper >= bound) {
      growth = upper - bound;
      if (fwd == 0u) cfv_set_heap_bound(sh, s, upper); else cfv_set_aux_bound(sh, s, upper);
    }
  }
  const u32 growth_cost = growth * K.memory_growth_ergs_per_byte;
  u32 after_growth;
  if (remaining_ergs >= growth_cost) {
    after_growth = remaining_ergs - growth_cost;
  } else {
    exceptions |= 8u;
    after_growth = 0;
  }
  const u32 decommit_cost = K.ergs_per_code_word_decommittment * code_len_words;  // :423-424
  u32 after_decommit;
  if (after_growth >= decommit_cost) {
    after_decommit = after_growth - decommit_cost;
  } else {
    exceptions |= 4u;
    after_decommit = after_growth;
  }
  u32 mapped_code_page = K.unmapped_page, mapped_blob = 0;  // UNMAPPED_PAGE (:162,439)
  ZKW_DIV_IF(exceptions) {  // :435-439
    s.flags |= FLAG_PENDING;
  } else {  // :441-455 decommit (helpers.rs:164-194 + SimpleDecommitter decommitter.rs:32-98)
    // known_hashes.get(&hash) (decommitter.rs:52-56): open addressing over the code hashes (built at upload, zkw_pre_hash),
    // normally one probe = one index load + the two wide loads of the candidate's hash
    u32 pre = 0xffffffffu;
    if (P.n_preimages) {
      u32 slot = zkw_pre_hash(code_hash.w) & P.pre_mask;
      for (u32 probe = 0; probe <= P.pre_mask; probe++, slot = (slot + 1u) & P.pre_mask) {
        const u32 cand = P.pre_index[slot];
        if (cand == 0) break;
        const uint4* h4 = (const uint4*)P.preimages[cand - 1u].hash;
        const uint4 h0 = h4[0], h1 = h4[1];
        const u32 diff = (h0.x ^ code_hash.w[0]) | (h0.y ^ code_hash.w[1]) | (h0.z ^ code_hash.w[2]) | (h0.w ^ code_hash.w[3]) | (h1.x ^ code_hash.w[4]) |
                         (h1.y ^ code_hash.w[5]) | (h1.z ^ code_hash.w[6]) | (h1.w ^ code_hash.w[7]);
        if (diff == 0) {
          pre = cand - 1u;
          break;
        }
      }
    }
    if (pre == 0xffffffffu) {  // decommitter.rs:54-56: Err propagates out of cycle()
      lane_fail(s, ZKW_STATUS_UNKNOWN_CODE_HASH);
      return;
    }
    // history.get(&hash) (decommitter.rs:58-79): the row of this (hash -> blob) pair — no scan, no capacity
    zkw_dev_history* hist = P.history + (u64)lane_inst(sh, s) * P.hist_pitch + pre;
    const zkw_dev_history seen = *hist;
    const bool fresh = seen.valid == 0;
    const u32 page = fresh ? candidate_page : seen.page;
    const u32 blob = P.preimages[pre].blob;
    const u32 blob_len = P.blob_dir[blob].y & 0xffffu;  // `values.len() as u16`
    if (fresh) {
      hist->valid = 1u;
      hist->page = page;
      CF(sh, s, CF_N_HISTORY)++;
    } else {
      after_decommit += decommit_cost;  // :450-453 refund
    }
    ZKW_STAMP(58)  // far call: exceptions, preimage + history lookup
    uint4* a = aux_alloc(P, sh, s, ZKW_AUX_DECOMMIT, fresh ? 1u : 0u, s.timestamp + 1, page, blob_len | (blob << 16));
    if (a) {
      a[1] = u256_lo4(code_hash);
      a[2] = u256_hi4(code_hash);
      a[3] = make_uint4(pre, 0, 0, 0);  // preimage index: selects the cached sponge midstate of this code hash (zkw_commit.hip)
    }
#ifdef ZKW_WIDE
    if (a && P.commit_out && (sh.debug_flags & 16u) && (sh.debug_flags & ZKW_DQ_HELPER)) {
      // The workgroup has a helper wave (zkw_dq_helper): the decommit is handed over through LDS instead of being
      // chained here, where its permutation (~80k clocks for a lone wave) sits on this wave's critical path.  The record
      // goes into the slot the wave posts next; it becomes valid at the end of the cycle, if the cycle completes.
      const u32 area = dq_helper_area(sh);
      u32 posted = *ZKW_LDS_AT(area);
      while (posted - *ZKW_LDS_AT(area + 4u) >= 2u) ZKW_SLEEP(8);  // both slots still with the helper
      const u32 row = area + 16u + (posted & 1u) * 768u + zkw_lane_id() * 4u;
      *ZKW_LDS_AT(row) = pre | ((fresh ? 1u : 0u) << 30);  // (bit 31 = valid: set at the end of the cycle)
      *ZKW_LDS_AT(row + 256u) = s.timestamp + 1;
      *ZKW_LDS_AT(row + 512u) = page;
      s.kflags |= KF_DQ_CHAINED;
    } else
#endif
    if (a && P.commit_out && (sh.debug_flags & 16u)) {  // requested per launch (zkw_batches_step with the decommit queue in its mask)
      // decommit-queue commitment, chained here (zkw_commit.hip spec: leaf = sponge(code hash | length | blob digest),
      // cached per preimage at upload; tail' = P(leaf | tail | index | queue | timestamp, fresh | page)): one permutation
      // per decommit, in the shadow of a far call — instead of a bucket pass and a chain kernel over the aux stream
      // after the run
      const u32 inst = lane_inst(sh, s);
      u64* tail_p = P.commit_out + ((u64)inst * ZKW_QUEUE_COUNT + ZKW_QUEUE_DECOMMIT) * 4;
      const u32 j = P.dq_count[inst];
      const u64* ms = P.midstates + (u64)pre * 12;  // the leaf of this code (zkw_midstate_kernel)
      const u64 leaf[4] = {ms[0], ms[1], ms[2], ms[3]};
      u64 tail[4] = {tail_p[0], tail_p[1], tail_p[2], tail_p[3]};
      // A cycle can still fail behind this point (no arena slot / callstack depth / aux or register-delta capacity left:
      // ZKW_STATUS_LIMIT), and a failed cycle leaves no records — the chains computed from the streams after a run never
      // see this decommit.  The previous tail is kept so that a lane that leaves the cycle loop failed takes the step back
      // (dq_undo, at the loop exit).
      u64* prev_p = P.dq_prev + (u64)inst * 4;
      prev_p[0] = tail[0]; prev_p[1] = tail[1]; prev_p[2] = tail[2]; prev_p[3] = tail[3];
      gl_chain_step(P.commit_rc, leaf, tail, (u64)j + 1, ZKW_QUEUE_DECOMMIT, (u64)(s.timestamp + 1) | ((u64)(fresh ? 1u : 0u) << 32), (u64)page);
      tail_p[0] = tail[0]; tail_p[1] = tail[1]; tail_p[2] = tail[2]; tail_p[3] = tail[3];
      P.dq_count[inst] = j + 1;
      s.kflags |= KF_DQ_CHAINED;
    }
    ZKW_STAMP(59)  // far call: decommit event + chained commitment
    mapped_code_page = page;
    mapped_blob = blob;
  }
  // :468-487
  const u32 max_passable = (after_decommit / 64u) * 63u;
  const u32 leftover = after_decommit - max_passable;
  u32 passed, remaining_here;
  if (max_passable < abi_ergs) {
    passed = max_passable;
    remaining_here = leftover;
  } else {
    passed = abi_ergs;
    remaining_here = leftover + (max_passable - abi_ergs);
  }
  s.ergs = remaining_here;  // :490-495
  s.pc = ps.new_pc;
  prev[E_SP_PC] = (s.sp & 0xffffu) | (s.pc << 16);
  prev[E_ERGS] = s.ergs;
  prev[E_HEAP_BOUND] = cfv_heap_bound(sh, s);
  prev[E_AUX_BOUND] = cfv_aux_bound(sh, s);
  const u32 new_static = ((s.kflags & KF_STATIC) != 0 || is_static_call) ? 1u : 0u;
  CF(sh, s, CF_MPC) += K.new_memory_pages_per_far_call;  // :503
  s.kflags |= KF_COLD_DIRTY;
  // r15 = CALL_IMPLICIT_PARAMETER_REG_IDX :506-508 (read by the caller)
  u32 next[32];
#pragma unroll
  for (int i = 0; i < 32; i++) next[i] = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    u32 this_a, sender_a;
    if (variant == ZKW_FAR_NORMAL) {
      this_a = called[i];
      sender_a = prev[E_THIS + i];
    } else if (variant == ZKW_FAR_DELEGATE) {
      this_a = prev[E_THIS + i];
      sender_a = prev[E_SENDER + i];
    } else {
      this_a = called[i];
      sender_a = r15.w[i];
    }
    next[E_THIS + i] = this_a;
    next[E_SENDER + i] = sender_a;
    next[E_CODE_ADDR + i] = called[i];
  }
  next[E_BASE_PAGE] = new_base;
  next[E_CODE_PAGE] = mapped_code_page;
  next[E_SP_PC] = K.initial_sp_on_far_call > 0xffffu ? 0xffffu : K.initial_sp_on_far_call;  // pc = 0
  next[E_EH_FLAGS] = (handler & 0xffffu) | (new_static << 16);                               // is_local_frame = false
  next[E_ERGS] = passed;
  next[E_SHARDS] = new_this_shard | (caller_shard << 8) | (new_code_shard << 16);
#pragma unroll
  for (int i = 0; i < 4; i++) next[E_CTX + i] = variant == ZKW_FAR_DELEGATE ? prev[E_CTX + i] : CF(sh, s, CF_CTX0 + i);
  next[E_HEAP_BOUND] = K.new_frame_memory_stipend;
  next[E_AUX_BOUND] = K.new_frame_memory_stipend;
  next[E_CODE_BLOB] = mapped_blob;
  CF(sh, s, CF_CTX0 + 0) = CF(sh, s, CF_CTX0 + 1) = CF(sh, s, CF_CTX0 + 2) = CF(sh, s, CF_CTX0 + 3) = 0;  // :558
  // memory.start_global_frame (memory.rs:573-657): a fresh arena slot, pages lazily zero
  u32 new_slot = CF(sh, s, CF_NEXT_SLOT);
  if (new_slot < P.F) {
    CF(sh, s, CF_NEXT_SLOT)++;  // a fresh slot
  } else {  // reuse: a slot whose frame returned (the reference's page pools, memory.rs:15-148), else one that only `dump_page_content` still sees
    const zkw_dev_frame_meta* fms = frame_metas(P, sh, s);
    u32 dead = 0xffffffffu;
    new_slot = 0xffffffffu;
    for (u32 i = 0; i < P.F; i++) {
      const u32 st = fms[i].stack_hwm;
      if (st == ZKW_SLOT_FREE) {
        if (new_slot == 0xffffffffu) new_slot = i;
      } else if (ZKW_SLOT_STATE(st) == ZKW_SLOT_DEAD && dead == 0xffffffffu) {
        dead = i;
      }
    }
    if (new_slot == 0xffffffffu) new_slot = dead;
    if (new_slot == 0xffffffffu) {
      lane_fail(s, ZKW_STATUS_LIMIT);
      return;
    }
  }
  next[E_SLOT] = new_slot;
  {
    zkw_dev_frame_meta* fm = frame_metas(P, sh, s) + new_slot;
    fm->base_page = new_base;
    fm->stack_hwm = 0;
    fm->heap_hwm = 0;
    fm->aux_hwm = 0;
  }
  ZKW_STAMP(60)  // far call: next-frame image
  start_frame(P, sh, s, prev, next, true);  // :562
  ZKW_STAMP(61)  // far call: start_frame
  if (!lane_ok(s)) return;
  // registers :573-610 (applied by the caller)
  out.v1 = fat_ptr_to_u256(abi);
  out.v2 = u256_zero();
  out.v2.w[0] = (constructor_call ? 1u : 0u) | (to_system ? 2u : 0u);
  out.action = ZKW_ACT_FAR | (to_system ? ZKW_ACT_TO_SYSTEM : 0u);
}

// ret.rs:9-265
ZD void op_ret(ZKW_KP P, Shared& sh, Lane& s, const Decoded& d, const Pre& ps, HeavyOut& out) {
  ZKW_DIV_SCOPE;
  const zkw_isa_consts ZKW_CONST_AS& K = P.consts;
  u32 variant = ZKW_ATTR_VARIANT(d.attr);
  s.flags &= FLAG_PENDING;  // :27
  u256 src0 = ps.src0;
  bool src0_ptr = ps.src0_ptr;
  if (variant == ZKW_RET_PANIC) {
    src0 = u256_zero();
    src0_ptr = false;
  }
  FatPtr ptr = fat_ptr_from(src0);
  const u32 fwd = forward_type(K.forwarding_codes, src0.w[7] & 0xffu);
  bool to_label = ZKW_ATTR_FLAGS(d.attr) & 1u;
  const u32 label_pc = d.imm0;
  u32 pve = 0;
  const bool local = (s.kflags & KF_LOCAL) != 0;
  if (!local) {  // :58-96
    if (fwd == 1u) {
      if (!src0_ptr) variant = ZKW_RET_PANIC;
      if (ptr.page < CF(sh, s, CF_BASE_PAGE)) variant = ZKW_RET_PANIC;
    }
    pve = fat_ptr_validate(ptr, fwd != 1u);
    if (pve) variant = ZKW_RET_PANIC;
    if (!(ptr.offset <= ptr.length)) variant = ZKW_RET_PANIC;
    if (variant == ZKW_RET_PANIC) ptr.offset = ptr.page = ptr.start = ptr.length = 0;
  }
  u32 ergs_remaining = s.ergs;
  if (!local) {  // :101-190
    if (variant == ZKW_RET_OK || variant == ZKW_RET_REVERT) {
      if (fwd == 1u) {
        ptr.start = ptr.start + ptr.offset;
        ptr.length = ptr.length - ptr.offset;
        ptr.offset = 0;
      } else if (fwd == 0u) {
        ptr.page = CF(sh, s, CF_BASE_PAGE) + 2;
      } else {
        ptr.page = CF(sh, s, CF_BASE_PAGE) + 3;
      }
    }
    u32 growth = 0;
    if (fwd != 1u) {
      u32 upper = ptr.start + ptr.length;
      if (pve & FPV_DEREF_BEYOND) upper = 0xffffffffu;
      const u32 bound = fwd == 0u ? cfv_heap_bound(sh, s) : cfv_aux_bound(sh, s);
      growth = upper < bound ? 0u : upper - bound;
    }
    const u32 cost = growth * K.memory_growth_ergs_per_byte;
    if (ergs_remaining >= cost) {
      ergs_remaining -= cost;
    } else {
      ergs_remaining = 0;
      variant = ZKW_RET_PANIC;
      ptr.offset = ptr.page = ptr.start = ptr.length = 0;
    }
  }
  const bool panicked = variant == ZKW_RET_REVERT || variant == ZKW_RET_PANIC;  // :196
  // finish_frame (helpers.rs:248-264)
  const u32* fin = (const u32*)entry_ptr(P, sh, s, s.depth);
  const u32 fin_eh = fin[E_EH_FLAGS] & 0xffffu;
  const u32 fin_mark = fin[E_JOURNAL_MARK];
  const u32 fin_heap_bound = cfv_heap_bound(sh, s), fin_aux_bound = cfv_aux_bound(sh, s);
  storage_finish_frame(P, sh, s, fin_mark, panicked);
  {
    uint4* a = aux_alloc(P, sh, s, ZKW_AUX_FRAME_FINISH, panicked ? 1u : 0u, 0, 0, 0);
    (void)a;  // header only
  }
  if (s.depth == 0) {  // pop on an empty callstack: unwrap() panic (execution_stack.rs:112)
    lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
    return;
  }
  ZKW_STAMP(62)  // ret: validation, finish_frame
  if (!local) hwm_writeback(P, sh, s);  // (a near-call frame shares the slot and the marks of the frame it returns to)
  const u32 fin_slot = cfv_slot(sh, s), fin_base = CF(sh, s, CF_BASE_PAGE);
  s.depth--;
  frame_load(P, sh, s, local);
  to_label = to_label && local;  // :202
  if (!local) {  // memory.finish_global_frame (memory.rs:660-758): see "Arena slots and page lifetimes" above
    zkw_dev_frame_meta* fms = frame_metas(P, sh, s);
    const u32 parent = cfv_slot(sh, s), rp = ptr.page, n_slots = CF(sh, s, CF_NEXT_SLOT);
    for (u32 k2 = 0; k2 < n_slots; k2++) {  // returndata pages this frame had received: forwarded upwards or out of reach (:725-756)
      const u32 st = fms[k2].stack_hwm;
      if (ZKW_SLOT_STATE(st) != ZKW_SLOT_KEPT || st == ZKW_SLOT_FREE || (st & 0xffffu) != fin_slot) continue;
      const u32 kind = (st >> 16) & 3u;
      fms[k2].stack_hwm = (rp != 0 && rp == fms[k2].base_page + kind) ? (ZKW_SLOT_KEPT | (kind << 16) | parent) : (ZKW_SLOT_DEAD | (kind << 16));
    }
    if (rp != 0 && rp == fin_base + 2u) {         // the heap is the returndata (:702-715)
      fms[fin_slot].stack_hwm = ZKW_SLOT_KEPT | (2u << 16) | parent;
      fms[fin_slot].aux_hwm = 0;
    } else if (rp != 0 && rp == fin_base + 3u) {  // the aux heap is (:716-731)
      fms[fin_slot].stack_hwm = ZKW_SLOT_KEPT | (3u << 16) | parent;
      fms[fin_slot].heap_hwm = 0;
    } else {                                      // forwarding or a panic: every page of the frame goes back to the pool (:732-750)
      fms[fin_slot].stack_hwm = ZKW_SLOT_FREE;
      fms[fin_slot].base_page = 0;
    }
    out.v1 = fat_ptr_to_u256(ptr);
    out.action = ZKW_ACT_RET;
    if (CF(sh, s, CF_CTX0 + 0) | CF(sh, s, CF_CTX0 + 1) | CF(sh, s, CF_CTX0 + 2) | CF(sh, s, CF_CTX0 + 3)) s.kflags |= KF_COLD_DIRTY;
    CF(sh, s, CF_CTX0 + 0) = CF(sh, s, CF_CTX0 + 1) = CF(sh, s, CF_CTX0 + 2) = CF(sh, s, CF_CTX0 + 3) = 0;
  }
  s.ergs += ergs_remaining;  // :243
  if (to_label) s.pc = label_pc;
  else if (panicked) s.pc = fin_eh;
  if (local) {  // :254-260
    if (fin_heap_bound < cfv_heap_bound(sh, s) || fin_aux_bound < cfv_aux_bound(sh, s)) {
      lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
      return;
    }
    cfv_set_heap_bound(sh, s, fin_heap_bound);
    cfv_set_aux_bound(sh, s, fin_aux_bound);
  }
  if (variant == ZKW_RET_PANIC) s.flags |= FLAG_LT;  // :262-264
}

// ---------------------------------------------------------------------------------------------
// precompiles: see zkw_precompiles.hip.h (keccak256 / sha256 round functions over the lane's memory)
// ---------------------------------------------------------------------------------------------
// wave-uniform value that reached a function through a (vector) argument register: back into a scalar register
ZD u32 zkw_uniform(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }

#include "zkw_precompiles.hip.h"


// The precompile bodies (Keccak-f state of 50 VGPRs, SHA-256 schedule, secp256k1) are compiled as ONE out-of-line
// function.  Everything crosses the call in the 32 vector argument registers — a: the lane state (lane_pack) + [12] which
// precompile, [13] the query's timestamp; b: the eight dwords of the call's ABI word (PrecompileCallABI: offsets, lengths,
// pages) — and the lane state comes back the same way; the wave's Shared view is rebuilt from the wave's LDS header as in
// zkw_heavy_body.  (Round 4 passed the Lane and a whole LogQuery by value: 49 dwords, i.e. through scratch memory — 608 bytes
// of frame in the caller and 336 here, on the call chain that sizes the cycle kernel's private segment.)
#ifndef ZKW_EMU_BUILD
typedef u32 zkw_v16 __attribute__((ext_vector_type(16)));
#else  // g++ (tests/emu) has no ext_vector_type
struct zkw_v16 {
  u32 v[16];
  u32& operator[](int i) { return v[i]; }
  const u32& operator[](int i) const { return v[i]; }
};
#endif
ZD zkw_v16 lane_pack(const Lane& s) {
  zkw_v16 a;
  a[0] = s.pc; a[1] = s.sp; a[2] = s.ergs; a[3] = s.timestamp; a[4] = s.prev_super_pc; a[5] = s.depth; a[6] = s.status; a[7] = s.flags;
  a[8] = s.kflags; a[9] = s.ptr_bitmap; a[10] = s.reg_dirty; a[11] = s.counts; a[12] = 0; a[13] = 0; a[14] = 0; a[15] = 0;
  return a;
}
ZD void lane_unpack(Lane& s, const zkw_v16& a) {
  s.pc = a[0]; s.sp = a[1]; s.ergs = a[2]; s.timestamp = a[3]; s.prev_super_pc = a[4]; s.depth = a[5]; s.status = a[6]; s.flags = a[7];
  s.kflags = a[8]; s.ptr_bitmap = a[9]; s.reg_dirty = a[10]; s.counts = a[11];
}
template <int WHICH>
ZD zkw_v16 zkw_precompile_body(zkw_v16 a, zkw_v16 b) {
  const u32 wib = zkw_uniform(threadIdx.x / ZKW_WAVE);
  const uint4 hdr = *(zkw_lds + ZKW_LDS_WAVES0 + wib * zkw_wave_lds_units() + 1);  // written by the kernel prologue
  const zkw_kparams ZKW_CONST_AS* Pp = (const zkw_kparams ZKW_CONST_AS*)(((u64)zkw_uniform(hdr.y) << 32) | zkw_uniform(hdr.x));
  ZKW_KP P = *Pp;
  Shared sh;
  shared_setup(sh, P, zkw_uniform(hdr.z), wib, zkw_uniform(hdr.w), false);
  Lane s;
  s.lane = zkw_lane_id();
  lane_unpack(s, a);
  LogQ q;  // (the precompiles read the ABI word and the timestamp only)
#pragma unroll
  for (int i = 0; i < 8; i++) q.key.w[i] = b[i];
  q.read_value = u256_zero(); q.written_value = u256_zero();
#pragma unroll
  for (int i = 0; i < 5; i++) q.address[i] = 0;
  q.timestamp = a[13]; q.tx_number = 0; q.aux_byte = 0; q.shard_id = 0;
  q.rw = false; q.rollback = false; q.is_service = false;
  if (WHICH == 0) precompile_keccak256(P, sh, s, q);
  else if (WHICH == 1) precompile_sha256(P, sh, s, q);
  else precompile_ecrecover(P, sh, s, q);
  s.lane = zkw_lane_id();
  return lane_pack(s);
}
// (one function per precompile: each has the frame IT needs — as one function the frame was the three side by side)
static __device__ __noinline__ zkw_v16 zkw_precompile_keccak(zkw_v16 a, zkw_v16 b) { return zkw_precompile_body<0>(a, b); }
static __device__ __noinline__ zkw_v16 zkw_precompile_sha(zkw_v16 a, zkw_v16 b) { return zkw_precompile_body<1>(a, b); }
static __device__ __noinline__ zkw_v16 zkw_precompile_ec(zkw_v16 a, zkw_v16 b) { return zkw_precompile_body<2>(a, b); }

#ifdef ZKW_WIDE
// `count` consecutive positions of the memory-query stream for every active lane (stream_alloc: one).  A lane whose
// positions would not fit below `cap` takes none and gets ~0.
ZD u32 stream_alloc_counted(u32 count, u32 cap) {
  const u32 base = zkw_cursor_get<0>();
  u32 total = 0, my = 0xffffffffu;
  for (u64 m = zkw_ballot(1); m; m &= m - 1) {
    const u32 l = (u32)__builtin_ctzll(m);
    const u32 c = (u32)__builtin_amdgcn_readlane((int)count, (int)l);
    const bool fits = c <= cap && base + total <= cap - c;
    if (zkw_lane_id() == l && fits) my = base + total;
    if (fits) total += c;
  }
  zkw_cursor_set<0>(base + total);
  return my;
}

// keccak256 of a wave that has a helper (ZKW_KECCAK_HELPER, zkw_kh_helper): what precompile_keccak256 does, with the
// reads, their witness and the sponge handed over.  The requester resolves the page (the unreachable-page status is its
// own), allocates the stream positions and the sequence numbers of the input reads — its later records then follow
// them, in the order the reference emits — posts the request in its row of the mailbox and waits for the digest.
ZD void keccak_request(ZKW_KP P, Shared& sh, Lane& s, const LogQ& q) {
  ZKW_DIV_SCOPE;  // (a lane that fails leaves early; the others wait for their digests and write them)
  const u32 in_off = q.key.w[0], in_len = q.key.w[1], out_off = q.key.w[2];
  const u32 page_r = q.key.w[4], page_w = q.key.w[5];
  const u32 n_words = in_len ? ((in_off & 31u) + in_len + 31u) >> 5 : 0u;
  FatPage fp;
  fp.slot = 0; fp.hwm = 0; fp.found = true; fp.is_aux = false; fp.empty = true;
  if (n_words) fp = fat_ptr_resolve(P, sh, s, page_r);
  const u32 pos = stream_alloc_counted(lane_ok(s) ? n_words : 0u, sh.cap_mem);
  if (!lane_ok(s)) return;
  const u32 seq0 = s.counts & 255u, mem0 = (s.counts >> 8) & 255u;
  const u32 seq1 = n_words < 255u - seq0 ? seq0 + n_words : 255u, mem1 = n_words < 255u - mem0 ? mem0 + n_words : 255u;
  s.counts = (s.counts & 0xffff0000u) | seq1 | (mem1 << 8);
  if (pos == 0xffffffffu) {
    lane_fail(s, ZKW_STATUS_LIMIT);
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the heap words this wave has stored are what the helper reads
  const u32 row = kh_box(sh) + zkw_lane_id() * 64u;
#ifdef __HIP_DEVICE_COMPILE__
  ZKW_LDS_AS u32* rw = (ZKW_LDS_AS u32*)(size_t)row;
#else
  u32* rw = (u32*)((char*)zkw_lds + row);
#endif
  *ZKW_LDS_WORD(rw + 1) = in_off;
  *ZKW_LDS_WORD(rw + 2) = in_len;
  *ZKW_LDS_WORD(rw + 3) = fp.hwm;
  *ZKW_LDS_WORD(rw + 4) = pos;
  *ZKW_LDS_WORD(rw + 5) = seq0;
  *ZKW_LDS_WORD(rw + 6) = q.timestamp;
  *ZKW_LDS_WORD(rw + 7) = page_r;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  *ZKW_LDS_WORD(rw) = 0x80000000u | fp.slot | (fp.is_aux ? 1u << 16 : 0u) | (fp.empty ? 1u << 17 : 0u);
  {
    ZKW_DIV_SCOPE;  // (the digests of the lanes arrive one by one)
    while (*ZKW_LDS_WORD(rw) >> 31) ZKW_SLEEP(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  u256 digest;
#pragma unroll
  for (int i = 0; i < 8; i++) digest.w[i] = *ZKW_LDS_WORD(rw + 8 + i);
  heap_write_cur(P, sh, s, false, out_off, digest);
  emit_mem(P, sh, s, q.timestamp + 1, ZKW_MEM_HEAP, page_w, out_off, digest, false, true, 2);
}
#endif

// helpers.rs:196-223 + DefaultPrecompilesProcessor dispatch on the low 16 address bits
ZD void call_precompile(ZKW_KP P, Shared& sh, Lane& s, const LogQ& q) {
  emit_log(P, sh, s, q, ZKW_LQ_LOG);
  const u32 addr_low = q.address[0] & 0xffffu;
  u32 which = 3;
  if (addr_low == P.consts.keccak_precompile_address) which = 0;
  else if (addr_low == P.consts.sha256_precompile_address) which = 1;
  else if (addr_low == P.consts.ecrecover_precompile_address) which = 2;
#ifdef ZKW_WIDE
  // (lengths near 2^32 — no real call: the stream capacity ends it — keep the one path whose arithmetic on them the oracle
  // is checked against; the helper's block count must stay bounded by the positions the requester could allocate)
  ZKW_DIV_IF(which == 0 && (sh.debug_flags & ZKW_KECCAK_HELPER) && q.key.w[1] < 0x40000000u) {
    keccak_request(P, sh, s, q);
    which = 3;
  }
#endif
  // anything else behaves as an unknown precompile: no memory traffic.  The lanes of a group may call different
  // precompiles (the address is per lane): one call per kind present.
  for (u32 k = 0; k < 3; k++) {
    ZKW_DIV_IF(which == k) {
      zkw_v16 pa = lane_pack(s), pb;
      pa[12] = k; pa[13] = q.timestamp;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        pb[i] = q.key.w[i];
        pb[8 + i] = 0;
      }
      const zkw_v16 pr = k == 0 ? zkw_precompile_keccak(pa, pb) : (k == 1 ? zkw_precompile_sha(pa, pb) : zkw_precompile_ec(pa, pb));
      lane_unpack(s, pr);
      s.lane = zkw_lane_id();
    }
  }
}

// One out-of-line function for the heavy, rare opcode bodies: near_call, log (storage, events, precompile calls),
// far_call, ret.  Inlined into the cycle loop their live ranges (two 32-dword callstack images, a 40-dword LogQuery,
// the storage probe) would add to the per-lane state, and the kernel would not fit the 128 registers the compiler has.
// Everything crosses the call in vector registers (two 16-dword arguments, one 16-dword result: the AMDGPU calling
// convention passes 32 argument dwords in registers, anything more — and any struct result — goes through scratch
// memory, i.e. through HBM at this kernel's footprint) or in LDS: the parameter-block pointer and the debug flags sit in
// the wave's LDS header, one 256-bit value per lane (r15 in, the value for dst0 / r1 out) in sh.xfer.  All lanes of a
// call hold the same instruction word, so it is made scalar again on entry.
// a: lane state (lane_pack) + [12] opcode word low, [13] high, [14] packed ISA attributes | src0_ptr << 30 | src1_ptr << 31
// b: src0 (8 dwords), src1 (8 dwords)
// result: lane state + [12] action bits (ZKW_ACT_*), [13] low dword of the second value (far call: r2)
template <u32 OPCODE, int KIND>
ZD zkw_v16 zkw_heavy_body(zkw_v16 a, zkw_v16 b) {
  const u32 wib = zkw_uniform(threadIdx.x / ZKW_WAVE);
  const uint4 hdr = *(zkw_lds + ZKW_LDS_WAVES0 + wib * zkw_wave_lds_units() + 1);  // written by the kernel prologue
  const zkw_kparams ZKW_CONST_AS* Pp = (const zkw_kparams ZKW_CONST_AS*)(((u64)zkw_uniform(hdr.y) << 32) | zkw_uniform(hdr.x));
  ZKW_KP P = *Pp;
  Shared sh;
  shared_setup(sh, P, zkw_uniform(hdr.z), wib, zkw_uniform(hdr.w), false);
  ZKW_STAMP(48)  // marshalling, call, prologue, parameter block
  Lane s;
  s.lane = zkw_lane_id();
  lane_unpack(s, a);
  Decoded d;
  const u32 u_lo = zkw_uniform(a[12]), u_hi = zkw_uniform(a[13]);
  d.word_lo = u_lo; d.word_hi = u_hi;
  d.attr = zkw_uniform(a[14] & 0x3fffffffu);
  d.cond = (u_lo >> 13) & 7u; d.src0 = (u_lo >> 16) & 15u; d.src1 = (u_lo >> 20) & 15u; d.dst0 = (u_lo >> 24) & 15u; d.dst1 = u_lo >> 28;
  d.imm0 = u_hi & 0xffffu; d.imm1 = u_hi >> 16;
  Pre ps;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    ps.src0.w[i] = b[i];
    ps.src1.w[i] = b[8 + i];
  }
  ps.src0_ptr = (a[14] >> 30) & 1u;
  ps.src1_ptr = (a[14] >> 31) & 1u;
  ps.dst0.has_loc = false; ps.dst0.type = 0; ps.dst0.page = 0; ps.dst0.index = 0;  // the destination is the caller's business
  ps.new_pc = (s.pc + 1) & 0xffffu;
  HeavyOut out;
  out.v1 = u256_zero();
  out.v2 = u256_zero();
  out.action = 0;
  const u32 opcode = OPCODE;  // one out-of-line function per opcode: each saves only the callee-saved registers IT uses
  ZKW_STAMP(49)
  if (opcode == ZKW_OP_LOG) {
    op_log<KIND>(P, sh, s, d, ps, out);
  } else if (opcode == ZKW_OP_NEAR_CALL) {
    op_near_call(P, sh, s, d, ps);
  } else if (opcode == ZKW_OP_FAR_CALL) {
    u256 r15;
#pragma unroll
    for (int i = 0; i < 8; i++) r15.w[i] = ZKW_XFER(sh, s, i);
    op_far_call(P, sh, s, d, ps, r15, out);
  } else {
    op_ret(P, sh, s, d, ps, out);
  }
  s.lane = zkw_lane_id();
  ZKW_STAMP(50 + (opcode == ZKW_OP_FAR_CALL ? 0 : opcode == ZKW_OP_RET ? 1 : opcode == ZKW_OP_NEAR_CALL ? 2 : 3))  // rest of the body
  if (out.action & (ZKW_ACT_DST0 | ZKW_ACT_FAR | ZKW_ACT_RET)) {
#pragma unroll
    for (int i = 0; i < 8; i++) ZKW_XFER(sh, s, i) = out.v1.w[i];
  }
  zkw_v16 r = lane_pack(s);
  r[12] = out.action;
  r[13] = out.v2.w[0];
  return r;
}

// (the call sequence of a function saves every callee-saved register the function touches — scalar ones through a
// memory round trip each when no vector register is free — so the four bodies do not share one function)
static __device__ __noinline__ zkw_v16 zkw_heavy_log_storage(zkw_v16 a, zkw_v16 b) { return zkw_heavy_body<ZKW_OP_LOG, 0>(a, b); }
static __device__ __noinline__ zkw_v16 zkw_heavy_log_event(zkw_v16 a, zkw_v16 b) { return zkw_heavy_body<ZKW_OP_LOG, 1>(a, b); }
static __device__ __noinline__ zkw_v16 zkw_heavy_log_precompile(zkw_v16 a, zkw_v16 b) { return zkw_heavy_body<ZKW_OP_LOG, 2>(a, b); }
static __device__ __noinline__ zkw_v16 zkw_heavy_near_call(zkw_v16 a, zkw_v16 b) { return zkw_heavy_body<ZKW_OP_NEAR_CALL, 0>(a, b); }
static __device__ __noinline__ zkw_v16 zkw_heavy_far_call(zkw_v16 a, zkw_v16 b) { return zkw_heavy_body<ZKW_OP_FAR_CALL, 0>(a, b); }
static __device__ __noinline__ zkw_v16 zkw_heavy_ret(zkw_v16 a, zkw_v16 b) { return zkw_heavy_body<ZKW_OP_RET, 0>(a, b); }
// A variant group's cycle (exec_decoded: `vec`): the lanes share the packed ISA entry (opcode, variant, addressing modes)
// but hold different register numbers and immediates, so the decode of those fields is per lane and the register file
// is reached through the waterfall accessors of RegFileVec.  a: lane state + [12] / [13] the lane's OWN instruction word,
// [14] the shared ISA entry, [15] = 1.  Result: the lane state.
template <class RF>
ZD void exec_decoded(ZKW_KP P, Shared& sh, RF& rf, Lane& s, const Decoded& d, bool vec, u32 vec_lo, u32 vec_hi);
static __device__ __noinline__ zkw_v16 zkw_vec_exec(zkw_v16 a, zkw_v16 b) {
#if defined(ZKW_WIDE)
  const u32 wib = zkw_uniform(threadIdx.x / ZKW_WAVE);
  const uint4 hdr = *(zkw_lds + ZKW_LDS_WAVES0 + wib * zkw_wave_lds_units() + 1);  // written by the kernel prologue
  const zkw_kparams ZKW_CONST_AS* Pp = (const zkw_kparams ZKW_CONST_AS*)(((u64)zkw_uniform(hdr.y) << 32) | zkw_uniform(hdr.x));
  ZKW_KP P = *Pp;
  Shared sh;
  shared_setup(sh, P, zkw_uniform(hdr.z), wib, zkw_uniform(hdr.w), false);
  Lane s;
  s.lane = zkw_lane_id();
  lane_unpack(s, a);
  const u32 lo = a[12], hi = a[13];  // per lane
  Decoded d;
  d.word_lo = lo; d.word_hi = hi;
  d.attr = zkw_uniform(a[14]);
  d.cond = (lo >> 13) & 7u; d.src0 = (lo >> 16) & 15u; d.src1 = (lo >> 20) & 15u; d.dst0 = (lo >> 24) & 15u; d.dst1 = lo >> 28;
  d.imm0 = hi & 0xffffu; d.imm1 = hi >> 16;
  RegFileVec rf;
#ifdef ZKW_EMU_BUILD
  rf.f = zkw_emu_rf_ptr[threadIdx.x];
#endif
  exec_decoded(P, sh, rf, s, d, false, 0u, 0u);
  s.lane = zkw_lane_id();
  return lane_pack(s);
#else
  (void)b;
  return a;  // never reached in the one-lane emulation build
#endif
}

// One call site in the cycle loop (several would change the register allocation of the whole loop): the dispatcher
// forwards its own arguments, so each branch is a tail call (a scalar jump; the opcode travels in a[14]).
static __device__ __noinline__ zkw_v16 zkw_heavy_entry(zkw_v16 a, zkw_v16 b) {
  if (zkw_uniform(a[15])) return zkw_vec_exec(a, b);  // a variant group, not a heavy body
  const u32 attr = zkw_uniform(a[14] & 0x3fffffffu);
  const u32 opcode = ZKW_ATTR_OPCODE(attr);
  if (opcode == ZKW_OP_LOG) {
    const u32 v = ZKW_ATTR_VARIANT(attr);
    if (v == ZKW_LOG_STORAGE_READ || v == ZKW_LOG_STORAGE_WRITE) return zkw_heavy_log_storage(a, b);
    if (v == ZKW_LOG_EVENT || v == ZKW_LOG_TO_L1) return zkw_heavy_log_event(a, b);
    return zkw_heavy_log_precompile(a, b);
  }
  if (opcode == ZKW_OP_NEAR_CALL) return zkw_heavy_near_call(a, b);
  if (opcode == ZKW_OP_FAR_CALL) return zkw_heavy_far_call(a, b);
  return zkw_heavy_ret(a, b);
}

// ---------------------------------------------------------------------------------------------
// read_and_decode exceptions (cycle.rs:142-184) and condition resolution (:193-209), branch-free
// ---------------------------------------------------------------------------------------------
ZD bool decode_exception(u32 max_depth, const Lane& s, u32 attr, u32 price) {
  const u32 props = ZKW_ATTR_PROPS(attr);
  return ((props & ZKW_PROP_EXPLICIT_PANIC) != 0) | (s.ergs < price) | (((props & ZKW_PROP_KERNEL_ONLY) != 0) & ((s.kflags & KF_KERNEL) == 0)) |
         (((props & ZKW_PROP_STATIC_OK) == 0) & ((s.kflags & KF_STATIC) != 0)) | (s.depth == max_depth);
}
// one bit per (condition, lt|eq<<1|gt<<2): Always, Gt, Lt, Eq, Ge, Le, Ne, GtOrLt
ZD bool condition_resolved(u64 lut, u32 cond, u32 flags) {  // lut = zkw_isa_consts.condition_lut
  return (lut >> (cond * 8 + (flags & 7u))) & 1ull;
}

// ---------------------------------------------------------------------------------------------
// operands .. opcode body of one cycle (cycle.rs:275-406) for one group of lanes that hold the SAME
// decoded instruction.  `d` is wave-uniform (its fields live in scalar registers), so every branch that
// depends on the opcode, the addressing modes or the register indices is a scalar branch; only the data
// path (256-bit values, sp, ergs, memory addresses) is per lane.
// ---------------------------------------------------------------------------------------------
// RF = RegFile: the scalar decode of a word group (register numbers are wave-uniform).  `vec` (wave-uniform) marks a VARIANT
// group instead — lanes that share the ISA entry but not the register numbers / immediates: their cycle runs out of line
// (zkw_vec_exec) in the RF = RegFileVec instantiation of this function, where `d`'s operand fields are per-lane values and
// the register file is reached through a waterfall.  Both out-of-line paths leave through ONE call (the heavy bodies and
// the variant groups share the dispatcher zkw_heavy_entry): a second call site changes the register allocation of the
// whole cycle loop.
template <class RF>
ZD void exec_decoded(ZKW_KP P, Shared& sh, RF& rf, Lane& s, const Decoded& d, bool vec, u32 vec_lo, u32 vec_hi) {
  constexpr bool IS_VEC = ZKW_RF_IS_VEC(RF);
  const u32 opcode = ZKW_ATTR_OPCODE(d.attr);
  const u32 props = ZKW_ATTR_PROPS(d.attr);
  const bool set_flags = ZKW_ATTR_FLAGS(d.attr) & 1u;
  ZKW_SUB_DECL
  s.lane = zkw_lane_id();
  // ----------------------------------------------------------------------------------------
  // operands (cycle.rs:275-350)
  // ----------------------------------------------------------------------------------------
  Pre ps;
  bool call_out = false;  // this lane leaves through the out-of-line call below
  // (the call's argument registers are filled at the call site, behind a SCALAR test of the opcode: as values defined up
  // here they were live — and zeroed, 32 moves — on the path of every light opcode)
  if (ZKW_UNLIKELY(!IS_VEC && vec)) {  // wave-uniform
    call_out = true;
  } else {
  u32 sp = s.sp;
  // all register-file reads of the cycle are issued back to back (one LDS round trip instead of three)
  u32 src0_reg_ptr, dummy_ptr;
  const u256 src0_reg = reg_read(sh, rf, s, d.src0, src0_reg_ptr);
  ps.src1 = reg_read(sh, rf, s, d.src1, ps.src1_ptr);  // :339
  const u256 dst0_reg = ZKW_ATTR_DST0(d.attr) == ZKW_MODE_REG ? u256_zero() : reg_read(sh, rf, s, d.dst0, dummy_ptr);  // only addressing modes use it
  ZKW_SUB(67)  // operands: register reads
  Operand src0_loc = compute_address(P, sh, s, sp, src0_reg, d.imm0, ZKW_ATTR_SRC0(d.attr), false);
  ps.dst0 = compute_address(P, sh, s, sp, dst0_reg, d.imm1, ZKW_ATTR_DST0(d.attr), true);
  s.sp = sp;                                            // :297
  if (opcode == ZKW_OP_NOP) src0_loc.has_loc = false;  // :298-301
  u256 src0_mem = u256_zero();
  u32 src0_mem_ptr = 0;
  ZKW_SUB(68)  // operands: addresses
  if (ZKW_UNLIKELY(src0_loc.has_loc)) {  // :304-325
    if (src0_loc.type == ZKW_MEM_CODE) src0_mem = code_read(sh, s, src0_loc.index);
    else src0_mem = stack_read(P, sh, s, src0_loc.index, src0_mem_ptr);
    emit_mem(P, sh, s, s.timestamp, src0_loc.type, src0_loc.page, src0_loc.index, src0_mem, src0_mem_ptr, false, 0);
  }
  ZKW_SUB(69)  // operands: memory operand + its query
  const u32 src0_mode = ZKW_ATTR_SRC0(d.attr);
  if (src0_mode == ZKW_MODE_REG) {
    ps.src0 = src0_reg;
    ps.src0_ptr = src0_reg_ptr;
  } else if (src0_mode == ZKW_MODE_IMM) {
    ps.src0 = u256_from_u32(d.imm0);
    ps.src0_ptr = 0;
  } else {
    ps.src0 = src0_mem;
    ps.src0_ptr = src0_mem_ptr;
  }
  {  // swap_operands (:341-345) — limb-wise selects (a struct swap keeps both operands in scratch memory).  (Swapping the two
     // register NUMBERS instead when both operands are registers — scalars, free — was measured: fewer instructions, kernel
     // +1.6 % slower on the driver's command, profiles/r06_ab_log.txt)
    const bool sw = (props & ZKW_PROP_SWAP) != 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32 a = ps.src0.w[i], b = ps.src1.w[i];
      ps.src0.w[i] = sw ? b : a;
      ps.src1.w[i] = sw ? a : b;
    }
    const u32 ap = ps.src0_ptr, bp = ps.src1_ptr;
    ps.src0_ptr = sw ? bp : ap;
    ps.src1_ptr = sw ? ap : bp;
  }
  ps.new_pc = (s.pc + 1) & 0xffffu;  // :347-350 (never a skip cycle here)
  if (ZKW_UNLIKELY(!(s.kflags & KF_KERNEL))) {     // erase_fat_pointer_metadata :374-396
    if (!(props & ZKW_PROP_SRC0_PTR_OK) && ps.src0_ptr) {
      ps.src0.w[1] = 0;
      ps.src0.w[2] = 0;
      ps.src0_ptr = 0;
    }
    if (!(props & ZKW_PROP_SRC1_PTR_OK) && ps.src1_ptr) {
      ps.src1.w[1] = 0;
      ps.src1.w[2] = 0;
      ps.src1_ptr = 0;
    }
  }
  // ----------------------------------------------------------------------------------------
  // apply (opcodes/parsing.rs:47-79)
  // ----------------------------------------------------------------------------------------
  ZKW_SUB(40)  // operands
  ZKW_DIV_IF(ZKW_LIKELY(lane_ok(s))) {
    switch (opcode) {
      case ZKW_OP_NOP: s.pc = ps.new_pc; break;  // noop.rs:16-19
      case ZKW_OP_ADD:
      case ZKW_OP_SUB: {  // add.rs:35-53, sub.rs:35-54
        s.pc = ps.new_pc;
        bool of;
        const u256 r = opcode == ZKW_OP_ADD ? u256_add(ps.src0, ps.src1, of) : u256_sub(ps.src0, ps.src1, of);
        const bool eq = u256_is_zero(r);
        if (set_flags) set_flags3(s, of, eq, !eq && !of);
        dst0_update(P, sh, rf, s, ps.dst0, d.dst0, r, false);
        break;
      }
      case ZKW_OP_MUL: {  // mul.rs:35-65
        s.pc = ps.new_pc;
        u256 lo, hi;
        u256_mul(ps.src0, ps.src1, lo, hi);
        if (set_flags) {
          const bool of = !u256_is_zero(hi), eq = u256_is_zero(lo);
          set_flags3(s, of, eq, !of && !eq);
        }
        dst0_update(P, sh, rf, s, ps.dst0, d.dst0, lo, false);
        reg_write(sh, rf, s, d.dst1, hi, false);
        break;
      }
      case ZKW_OP_DIV: {  // div.rs:35-75
        s.pc = ps.new_pc;
        ZKW_DIV_IF(u256_is_zero(ps.src1)) {  // (the division holds a ballot: u256_divmod skips digit steps no lane needs)
          if (set_flags) set_flags3(s, true, false, false);
          dst0_update(P, sh, rf, s, ps.dst0, d.dst0, u256_zero(), false);
          reg_write(sh, rf, s, d.dst1, u256_zero(), false);
        } else {
          u256 q, r;
          u256_divmod(ps.src0, ps.src1, q, r);
          if (set_flags) set_flags3(s, false, u256_is_zero(q), u256_is_zero(r));
          dst0_update(P, sh, rf, s, ps.dst0, d.dst0, q, false);
          reg_write(sh, rf, s, d.dst1, r, false);
        }
        break;
      }
      case ZKW_OP_JUMP: s.pc = clip16(sh, ps.src0); break;  // jump.rs:23-25
      case ZKW_OP_SHIFT: {  // shift.rs:44-78
        s.pc = ps.new_pc;
        const u32 n = ps.src1.w[0] & 0xffu;
        const u32 v = ZKW_ATTR_VARIANT(d.attr);
        const bool cyclic = v == ZKW_SHIFT_ROL || v == ZKW_SHIFT_ROR;
        const bool right = v == ZKW_SHIFT_SHR || v == ZKW_SHIFT_ROR;
        u256 r;
        if (right) {
          r = u256_shr(ps.src0, n);
          if (cyclic) r = u256_or(r, u256_shl(ps.src0, 256u - n));
        } else {
          r = u256_shl(ps.src0, n);
          if (cyclic) r = u256_or(r, u256_shr(ps.src0, 256u - n));
        }
        if (set_flags) set_flags3(s, false, u256_is_zero(r), false);
        dst0_update(P, sh, rf, s, ps.dst0, d.dst0, r, false);
        break;
      }
      case ZKW_OP_BINOP: {  // binop.rs:42-61
        s.pc = ps.new_pc;
        const u32 v = ZKW_ATTR_VARIANT(d.attr);
        const u256 r = v == ZKW_BINOP_XOR ? u256_xor(ps.src0, ps.src1) : (v == ZKW_BINOP_AND ? u256_and(ps.src0, ps.src1) : u256_or(ps.src0, ps.src1));
        if (set_flags) set_flags3(s, false, u256_is_zero(r), false);
        dst0_update(P, sh, rf, s, ps.dst0, d.dst0, r, false);
        break;
      }
      case ZKW_OP_CONTEXT: op_context(P, sh, rf, s, d, ps); break;
      case ZKW_OP_PTR: op_ptr(P, sh, rf, s, d, ps); break;
      case ZKW_OP_LOG:
      case ZKW_OP_NEAR_CALL:
      case ZKW_OP_FAR_CALL:
      case ZKW_OP_RET: {  // out of line; what they write to registers comes back as actions
        if (IS_VEC) {  // (a variant group never holds a heavy opcode: see the group selection)
          lane_fail(s, ZKW_STATUS_REFERENCE_PANIC);
          break;
        }
        ZKW_STAMP0
        if (opcode == ZKW_OP_FAR_CALL) {  // CALL_IMPLICIT_PARAMETER_REG_IDX, far_call.rs:506-508
          const u256 r15 = rf_get(rf, ((P.consts.call_regs >> 16) & 0xffu) + 1u);
#pragma unroll
          for (int i = 0; i < 8; i++) ZKW_XFER(sh, s, i) = r15.w[i];
        }
        call_out = true;
        break;
      }
      case ZKW_OP_UMA: op_uma(P, sh, rf, s, d, ps); break;
      default: lane_fail(s, ZKW_STATUS_REFERENCE_PANIC); break;  // Opcode::Invalid => unreachable!() parsing.rs:77
    }
  }
  }  // !vec
#if defined(__HIP_DEVICE_COMPILE__) || defined(ZKW_EMU_BUILD)
  if constexpr (!IS_VEC) {
  const bool out_of_line = vec || opcode == ZKW_OP_LOG || opcode == ZKW_OP_NEAR_CALL || opcode == ZKW_OP_FAR_CALL || opcode == ZKW_OP_RET;  // wave-uniform
  if (ZKW_UNLIKELY(out_of_line))
  ZKW_DIV_IF(ZKW_UNLIKELY(call_out)) {
    zkw_v16 oa = lane_pack(s), ob;
#pragma unroll
    for (int i = 0; i < 16; i++) ob[i] = 0;
    if (vec) {
      oa[12] = vec_lo; oa[13] = vec_hi; oa[14] = d.attr; oa[15] = 1u;
    } else {
      oa[12] = d.word_lo; oa[13] = d.word_hi; oa[14] = d.attr | ((ps.src0_ptr ? 1u : 0u) << 30) | ((ps.src1_ptr ? 1u : 0u) << 31);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        ob[i] = ps.src0.w[i];
        ob[8 + i] = ps.src1.w[i];
      }
    }
    const zkw_v16 r = zkw_heavy_entry(oa, ob);
#ifdef __HIP_DEVICE_COMPILE__
    {
      // The register allocator keeps the loop's shared "zero high half" (every u32 -> u64 extension pairs with it) in a
      // caller-saved register and reloads it from scratch behind this call — lazily, on the way to the join below, where
      // the pending reload would make EVERY group iteration wait for vmcnt(0), i.e. for the stores of its opcode body.
      // A zero extension right here pulls the reload in front of the settle.
      const u64 z = (u64)r[0];
      asm volatile("" : : "v"(z));
    }
#endif
    ZKW_SETTLE(4 /* out-of-line call */);
    lane_unpack(s, r);
    s.lane = zkw_lane_id();
    ZKW_STAMP(55)  // return + epilogue of the callee
    if (!vec) {
      const u32 action = r[12];
      u256 v1 = u256_zero();
      if (action & (ZKW_ACT_DST0 | ZKW_ACT_FAR | ZKW_ACT_RET)) {
#pragma unroll
        for (int i = 0; i < 8; i++) v1.w[i] = ZKW_XFER(sh, s, i);
      }
      ZKW_DIV_IF(action & ZKW_ACT_DST0) dst0_update(P, sh, rf, s, ps.dst0, d.dst0, v1, false);
      if (action & ZKW_ACT_FAR) {  // far_call.rs:573-610, in the reference's order; the conventions are table constants (indices into `registers`: r1 = 0)
        const u32 cr = P.consts.call_regs, rg = P.consts.call_ranges;
        reg_write(sh, rf, s, (cr & 0xffu) + 1u, v1, true);                            // CALL_IMPLICIT_CALLDATA_FAT_PTR_REGISTER
        reg_write(sh, rf, s, ((cr >> 8) & 0xffu) + 1u, u256_from_u32(r[13]), false);  // CALL_IMPLICIT_CONSTRUCTOR_MARKER_REGISTER
        for (u32 q = rg & 0xffu; q < ((rg >> 8) & 0xffu); q++) {                       // CALL_SYSTEM_ABI_REGISTERS
          if (action & ZKW_ACT_TO_SYSTEM) s.ptr_bitmap &= ~(1u << q);  // drop the pointer marker only
          else reg_write(sh, rf, s, q + 1u, u256_zero(), false);
        }
        for (u32 q = (rg >> 16) & 0xffu; q < (rg >> 24); q++) reg_write(sh, rf, s, q + 1u, u256_zero(), false);  // CALL_RESERVED_RANGE
        reg_write(sh, rf, s, ((cr >> 16) & 0xffu) + 1u, u256_zero(), false);          // CALL_IMPLICIT_PARAMETER_REG_IDX
      }
      if (action & ZKW_ACT_RET) {  // ret.rs:213-233
        const u32 rr = P.consts.ret_regs;
        reg_write(sh, rf, s, (rr & 0xffu) + 1u, v1, true);                            // RET_IMPLICIT_RETURNDATA_PARAMS_REGISTER
        reg_write(sh, rf, s, ((rr >> 8) & 0xffu) + 1u, u256_zero(), false);           // RET_RESERVED_REGISTER_0..2
        reg_write(sh, rf, s, ((rr >> 16) & 0xffu) + 1u, u256_zero(), false);
        reg_write(sh, rf, s, (rr >> 24) + 1u, u256_zero(), false);
        for (u32 q = (rr >> 24) + 1u; q < ZKW_REGISTERS_COUNT; q++) reg_write(sh, rf, s, q + 1u, u256_zero(), false);  // "ALL other registers are zeroed out"
      }
      ZKW_STAMP(56)  // actions
      ZKW_SETTLE(5 /* call actions */);  // (the operand descriptor is reloaded from scratch for the actions: not carried to the join either)
    }
  }
  }
#endif
}

// =============================================================================================
// the cycle kernel
// =============================================================================================
// Per-lane state write-back (a later run continues from it / the host reads the final state).  Called by a lane at the
// moment it LEAVES the cycle loop, from inside the loop: nothing of the lane state is then live after the loop, so the
// compiler does not have to keep a per-iteration copy of ~30 registers for the lanes that have already left
// (a loop with divergent exits preserves every live-out value of the exited lanes on each iteration).
// `completed` = cycles this lane completed in this launch
ZD void lane_writeback(ZKW_KP P, Shared& sh, const RegFile& rf, Lane& s, u32 completed) {
  const u32 tid = s.lane;
  if (s.status == ZKW_STATUS_RUNNING && s.depth == 0) s.status = ZKW_STATUS_ENDED;  // execution_has_ended() (mod.rs:96-98)
  frame_writeback(P, sh, s);
  // An instance that has ended has no current frame with pages: the root entry names the bootloader's arena slot, whose
  // meta op_ret has just settled (finish_global_frame, memory.rs:668-731: stack page and the page that is not the
  // returndata back to the pool) — the marks frame_load picked up before that must not go back over it.
  if (s.depth != 0) hwm_writeback(P, sh, s);
  zkw_dev_scalars sc;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint4 v = ZKW_SLOT_READ(sh, tid, i);
    sc.prev_code_word[2 * i] = v.x;
    sc.prev_code_word[2 * i + 1] = v.y;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) sc.ctx_u128_reg[i] = CF(sh, s, CF_CTX0 + i);
  sc.ptr_bitmap = s.ptr_bitmap; sc.flags = s.flags; sc.prev_code_page = (s.kflags & KF_CODE_PAGE_CHANGED) ? CF(sh, s, CF_PREV_CODE_PAGE) : CF(sh, s, CF_CODE_PAGE); sc.timestamp = s.timestamp;
  sc.cycle_counter = CF(sh, s, CF_CYCLE_COUNTER0) + completed; sc.spent_pubdata = CF(sh, s, CF_SPENT_PUBDATA); sc.memory_page_counter = CF(sh, s, CF_MPC);
  sc.absolute_execution_step = P.scalars0[lane_inst(sh, s)].absolute_execution_step /* never changed by a run */; sc.ergs_per_pubdata = CF(sh, s, CF_ERGS_PP); sc.tx_number = CF(sh, s, CF_TX_NUMBER);
  sc.prev_super_pc = s.prev_super_pc; sc.depth = s.depth; sc.status = s.status; sc.n_cycles = CF(sh, s, CF_N_CYCLES0) + completed; sc.first_dynamic_page = CF(sh, s, CF_FIRST_DYN);
  sc.n_initial_slots = CF(sh, s, CF_N_INITIAL_SLOTS); sc.next_slot = CF(sh, s, CF_NEXT_SLOT); sc.journal_len = CF(sh, s, CF_JOURNAL_LEN); sc.n_history = CF(sh, s, CF_N_HISTORY);
  sc.reserved[0] = 0;
  P.scalars[lane_inst(sh, s)] = sc;
  uint4* rg = P.regs + (u64)sh.wave * ZKW_REG_CHUNKS * sh.L + tid;
  for (u32 r = 0; r < ZKW_REGISTERS_COUNT; r++) {
    const u256 v = rf_get(rf, r + 1);
    rg[(u64)(2 * r) * sh.L] = u256_lo4(v);
    rg[(u64)(2 * r + 1) * sh.L] = u256_hi4(v);
  }
}

#ifdef ZKW_WIDE
// The helper wave of a workgroup (zkw_launch_args.helpers): chains the decommit-queue commitment for the cycle waves of
// its workgroup.  A far call that decommits posts (preimage, timestamp | fresh, page) per lane into a two-slot ring in
// LDS (op_far_call, end of cycle); this wave takes the slots in order — lane l serves lane l of the posting wave: one
// permutation per record, exactly the step the far call would have run itself (gl_chain_step on the instance's running
// tail) — and leaves when every cycle wave has said it is done and every slot is consumed.  The permutation is ~80k
// clocks for a lone wave, twice per 256 cycles of cfg 2: off the critical path of the cycle waves when a CU has a wave
// slot to spare (the driver's 1280 waves = 5 per CU), which is the only case the runtime launches helpers for.
// (the wave `gw` of the launch -> its batch and its wave of that batch; false: no such wave)
ZD bool zkw_find_wave(const zkw_launch_args& A, u32 gw, u32& b, u32& wave) {
  if (gw >= A.wave_base[A.n_batches]) return false;
  if (A.uniform_waves) {
    b = gw / A.uniform_waves;
    wave = gw - b * A.uniform_waves;
  } else {
    u32 lo = 0, hi = A.n_batches;
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (A.wave_base[mid] <= gw) lo = mid; else hi = mid;
    }
    b = lo;
    wave = gw - A.wave_base[lo];
  }
  return true;
}

// one look at the hand-over area of a cycle wave: 0 = nothing posted and the wave still cycles, 1 = one slot served,
// 2 = nothing posted and the wave has left its loop
ZD u32 zkw_dq_serve(ZKW_KP P, u32 wave, u32 area, u32& consumed, u32 tid) {
  const u32 posted = zkw_lds_get(area);
  if (consumed == posted) {
    if (!zkw_lds_get(area + 8u)) return 0;               // still cycling
    return zkw_lds_get(area) != posted ? 0u : 2u;        // (posted between the two reads)
  }
  const u32 row = area + 16u + (consumed & 1u) * 768u + tid * 4u;
  const u32 w0 = zkw_lds_get(row), ts = zkw_lds_get(row + 256u), page = zkw_lds_get(row + 512u);
  const u32 inst = wave * P.L + tid;
  if ((w0 >> 31) && tid < P.L && inst < P.n_instances) {
    const u32 pre = w0 & 0x3fffffffu, fresh = (w0 >> 30) & 1u;
    u64* tail_p = P.commit_out + ((u64)inst * ZKW_QUEUE_COUNT + ZKW_QUEUE_DECOMMIT) * 4;
    const u32 j = P.dq_count[inst];
    const u64* ms = P.midstates + (u64)pre * 12;
    const u64 leaf[4] = {ms[0], ms[1], ms[2], ms[3]};
    u64 tail[4] = {tail_p[0], tail_p[1], tail_p[2], tail_p[3]};
    gl_chain_step(P.commit_rc, leaf, tail, (u64)j + 1, ZKW_QUEUE_DECOMMIT, (u64)ts | ((u64)fresh << 32), (u64)page);
    tail_p[0] = tail[0]; tail_p[1] = tail[1]; tail_p[2] = tail[2]; tail_p[3] = tail[3];
    P.dq_count[inst] = j + 1;
    ZKW_EMU_COUNT(5);
  }
  zkw_lds_put(row, 0);  // the entry is empty again (a lane that has left its loop never rewrites it)
  consumed++;
  if (tid == 0) zkw_lds_put(area + 4u, consumed);  // the slot is free
  return 1;
}

ZD void zkw_dq_helper(const zkw_launch_args& A, u32 tid) {
  const u32 g = A.waves_per_group;
  const u32 area0 = ZKW_LDS_ADDR(zkw_lds + ZKW_LDS_WAVES0 + g * zkw_wave_lds_units());
  u32 consumed[ZKW_MAX_WAVES_PER_GROUP];
#pragma unroll
  for (int h = 0; h < ZKW_MAX_WAVES_PER_GROUP; h++) consumed[h] = 0;
  for (;;) {
    bool all_done = true;
#pragma unroll
    for (int h = 0; h < ZKW_MAX_WAVES_PER_GROUP; h++) {
      if ((u32)h >= g) continue;
      u32 b, wave;
      if (!zkw_find_wave(A, blockIdx.x * g + (u32)h, b, wave)) continue;
      ZKW_KP P = *(const zkw_kparams ZKW_CONST_AS*)A.kp[b];
      if (wave >= P.n_waves) continue;
      if (zkw_dq_serve(P, wave, area0 + (u32)h * ZKW_DQ_HELPER_BYTES, consumed[h], tid) != 2u) all_done = false;
    }
    if (all_done) break;
    ZKW_SLEEP(32);
  }
}

// ---- keccak256 served by helper waves, one lane per 32-bit half of a state word -------------------------------------
// A batch of thin waves (a few instances per wave: the shape of a latency-bound caller, BASELINE cfg 3's 512 instances)
// leaves most lanes of the chip idle while every instance walks its message one permutation after the other: ~15 us per
// Keccak-f[1600] with the 25-word state in the registers of one lane.  With ZKW_KECCAK_HELPER every cycle wave has helper
// waves in its workgroup, and a keccak256 call is a request in the cycle wave's mailbox in LDS (one 16-dword row per
// lane: call_precompile -> keccak_request).  A helper serves one request at a time: the low half of state word i = x + 5y
// in lane i, the high half in lane 32 + i, so that one ds_bpermute moves both halves of a word.  The cross-lane terms of
// a round take three dependent fetch steps (13 ds_bpermute, 13 ALU operations per lane; measured on the cfg-3 lone batch:
// a ds_bpermute costs ~13 clocks of issue, so that summing the neighbour columns in the lane itself — two steps, 21
// fetches — is slower, and so are DPP row shifts + v_permlane32_swap in place of the three fetches of theta) —
//   C[i]  = A[i] ^ A[i+5] ^ A[i+10] ^ A[i+15] ^ A[i+20]          (indices mod 25: the column of lane i)
//   E[i]  = A[i] ^ C[x-1] ^ rotl(C[x+1], 1)                      (theta; the rotation funnels the two halves of C[x+1])
//   A'[d] = T[src d] ^ (~T[src(x+1)] & T[src(x+2)]),  T[s] = rotl(E[s], rho[s]),  src(X, Y) = ((X + 3Y) % 5, X)
// (rho, pi, chi: the destination fetches both halves of the three words it needs and funnels them; which half is the
// high operand — rotations by 32 and more exchange them — is folded into the fetch addresses).  The message bytes of
// the next block are loaded while the current one is permuted, and the lanes that hold no state (25..31) read and
// witness the input words of the block (precompile read queries at the stream positions and with the sequence numbers
// the requester has allocated).  The digest goes back through the row.
__device__ const unsigned char ZKW_KH_RHO[32] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3, 10, 43, 25, 39, 41,
                                                 45, 15, 21, 8,  18, 2,  61, 56, 14, 0,  0, 0,  0,  0,  0,  0};
ZD u32 kh_bp(u32 byte_addr, u32 v) { return (u32)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v); }

ZD void zkw_kh_serve(ZKW_KP P, u32 wave, u32 box, u32 r, u32 tid) {
  const u32 h = tid >> 5, idx = tid & 31u;  // half of the word (0 = low), state index
  const bool in_state = idx < 25u;
  const u32 X = idx % 5u, Y = idx / 5u;
  const u32 row = box + r * 64u;
  const u32 f = zkw_lds_get(row), in_off = zkw_lds_get(row + 4u), in_len = zkw_lds_get(row + 8u);
  const u32 hwm = zkw_lds_get(row + 12u), pos = zkw_lds_get(row + 16u), seq0 = zkw_lds_get(row + 20u), ts = zkw_lds_get(row + 24u), page = zkw_lds_get(row + 28u);
  const u32 slot = f & 0xffffu;
  const bool is_aux = (f >> 16) & 1u, empty = (f >> 17) & 1u;
  const u32 L = P.L, words = is_aux ? P.A : P.H;
  const uint4* arena = is_aux ? P.aux_heap + (u64)wave * P.F * P.A * 2u * L : P.heap + (u64)wave * P.F * P.H * 2u * L;
  const u32 lim = empty ? 0u : (hwm < words ? hwm : words);  // words at and beyond read as zero (memory.rs:490-495)
  const u32 w0 = in_off >> 5, phase = in_off & 31u;
  const u32 n_words = in_len ? (phase + in_len + 31u) >> 5 : 0u;
  const u32 nb = in_len / ZKW_KECCAK_RATE + 1u;
  uint4* mem = P.mem_stream + (u64)wave * P.cap_mem * 3 + pos;
  const u32 meta = (ZKW_MEM_FAT_PTR & ZKW_MQ_TYPE_MASK) | (1u << ZKW_MQ_KIND_SHIFT);
  // fetch addresses (bytes: lane * 4); lanes outside the state fetch themselves
#define ZKW_KH_LANE(hh, i) ((in_state ? (hh) * 32u + (i) : tid) * 4u)
  const u32 a_c1 = ZKW_KH_LANE(h, (idx + 5u) % 25u), a_c2 = ZKW_KH_LANE(h, (idx + 10u) % 25u), a_c3 = ZKW_KH_LANE(h, (idx + 15u) % 25u), a_c4 = ZKW_KH_LANE(h, (idx + 20u) % 25u);
  const u32 xm1 = Y * 5u + (X + 4u) % 5u, xp1 = Y * 5u + (X + 1u) % 5u;
  const u32 a_xm1 = ZKW_KH_LANE(h, xm1), a_xp1_own = ZKW_KH_LANE(h, xp1), a_xp1_other = ZKW_KH_LANE(1u - h, xp1);
  // pi: destination (X, Y) takes the word of source x = (X + 3Y) % 5, y = X
  u32 a_first[3], a_second[3], shift[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const u32 xs = (X + (u32)k) % 5u;                       // the column of the destination this term stands for
    const u32 sidx = in_state ? (xs + 3u * Y) % 5u + 5u * xs : 0u;
    const u32 amount = ZKW_KH_RHO[sidx];
    const bool swap = amount >= 32u || amount == 0u;        // (0 is taken as 64: both halves exchanged, funnel shift 0)
    a_first[k] = ZKW_KH_LANE(swap ? 1u - h : h, sidx);      // high operand of the funnel
    a_second[k] = ZKW_KH_LANE(swap ? h : 1u - h, sidx);
    shift[k] = (32u - (amount & 31u)) & 31u;
  }
#undef ZKW_KH_LANE
  const u32 iota_lane = idx == 0 ? 0xffffffffu : 0u;
  const unsigned char* bytes = (const unsigned char*)arena;
  // this lane's four bytes of block b of the padded message, little-endian (lanes 0..16 of either half)
  auto block_dword = [&](u32 b) -> u32 {
    u32 m = 0;
    if (b >= nb || idx >= 17u) return m;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32 rel = b * ZKW_KECCAK_RATE + 8u * idx + 4u * h + (u32)i;  // byte of the padded message
      u32 v = 0;
      if (rel < in_len) {
        const u32 p = in_off + rel, wi = p >> 5, back = 31u - (p & 31u);  // words are big-endian: byte `back` of the little-endian value
        if (wi < lim) v = bytes[((u64)(2u * (slot * words + wi) * L + r + (back >> 4) * L)) * 16u + (back & 15u)];
      } else if (rel == in_len) {
        v = 0x01u;  // pad10*1 with the legacy domain byte
      }
      if (rel == nb * ZKW_KECCAK_RATE - 1u) v |= 0x80u;
      m |= v << (8 * i);
    }
    return m;
  };
  u32 st = 0;
  u32 m = block_dword(0);
  u32 witnessed = 0;  // input words read and witnessed so far
  for (u32 b = 0; b < nb; b++) {
    st ^= m;
    m = block_dword(b + 1u);  // (in flight during the permutation)
    {  // the words this block reaches into for the first time: one per spare lane
      const u32 end = (b + 1u) * ZKW_KECCAK_RATE < in_len ? (b + 1u) * ZKW_KECCAK_RATE : in_len;
      const u32 upto = in_len ? (phase + end + 31u) >> 5 : 0u;
      const u32 i = witnessed + (idx - 25u);
      if (h == 0 && idx >= 25u && i < upto) {
        const u32 wi = w0 + i;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = make_uint4(0, 0, 0, 0);
        if (wi < lim) {
          const uint4* e = arena + (2u * (slot * words + wi) * L + r);
          lo = zkw_gload4(e);
          hi = zkw_gload4(e + L);
        }
        const u32 seq = seq0 + i < 255u ? seq0 + i : 255u;
        zkw_stream_store(mem + i, make_uint4(ts, page, wi, r | (seq << 8) | (meta << 16)));
        zkw_stream_store(mem + i + P.cap_mem, lo);
        zkw_stream_store(mem + i + 2u * (u64)P.cap_mem, hi);
      }
      witnessed = upto;
    }
#pragma unroll
    for (int round = 0; round < 24; round++) {
      const u32 c = st ^ kh_bp(a_c1, st) ^ kh_bp(a_c2, st) ^ kh_bp(a_c3, st) ^ kh_bp(a_c4, st);
      const u32 e = st ^ kh_bp(a_xm1, c) ^ __builtin_amdgcn_alignbit(kh_bp(a_xp1_own, c), kh_bp(a_xp1_other, c), 31);
      const u32 t0 = __builtin_amdgcn_alignbit(kh_bp(a_first[0], e), kh_bp(a_second[0], e), shift[0]);
      const u32 t1 = __builtin_amdgcn_alignbit(kh_bp(a_first[1], e), kh_bp(a_second[1], e), shift[1]);
      const u32 t2 = __builtin_amdgcn_alignbit(kh_bp(a_first[2], e), kh_bp(a_second[2], e), shift[2]);
      const u32 rc = h ? (u32)(ZKW_KECCAK_RC[round] >> 32) : (u32)ZKW_KECCAK_RC[round];
      st = t0 ^ (~t1 & t2) ^ (iota_lane & rc);
    }
  }
  if (idx < 4u) zkw_lds_put(row + 32u + (7u - 2u * idx - h) * 4u, __builtin_bswap32(st));  // digest = state words 0..3, big-endian (as precompile_keccak256 writes it)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (tid == 0) ZKW_EMU_COUNT(4);
  if (tid == 0) zkw_lds_put(row, 0);  // served
}

// helper `sub` of the `n_sub` helpers of cycle wave h: requests of the rows r with r % n_sub == sub; the first also chains
// the wave's decommits (ZKW_DQ_HELPER)
ZD void zkw_kh_helper(const zkw_launch_args& A, u32 tid, u32 h, u32 sub, u32 n_sub) {
  const u32 g = A.waves_per_group;
  u32 b, wave;
  if (!zkw_find_wave(A, blockIdx.x * g + h, b, wave)) return;
  ZKW_KP P = *(const zkw_kparams ZKW_CONST_AS*)A.kp[b];
  if (wave >= P.n_waves) return;
  const u32 area0 = ZKW_LDS_ADDR(zkw_lds + ZKW_LDS_WAVES0 + g * zkw_wave_lds_units());
  const u32 area = area0 + h * ZKW_DQ_HELPER_BYTES, box = area0 + g * ZKW_DQ_HELPER_BYTES + h * ZKW_KH_BYTES;
  u32 consumed = 0;
  for (;;) {
    const u64 req = zkw_ballot(tid < ZKW_KH_MAX_LANES && tid % n_sub == sub && (zkw_lds_get(box + tid * 64u) >> 31));
    if (req) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the heap words the requester stored
      zkw_kh_serve(P, wave, box, (u32)__builtin_ctzll(req), tid);
      continue;
    }
    if (sub == 0) {
      const u32 st = zkw_dq_serve(P, wave, area, consumed, tid);
      if (st == 2u) break;  // (a wave that has left its loop has no request pending: the requester waits for its digest)
      if (st == 0u) ZKW_SLEEP(2);
    } else {
      if (zkw_lds_get(area + 8u)) break;
      ZKW_SLEEP(2);
    }
  }
}
#endif

// compiled for 128 vector registers (v0..v127); v128..v255 hold the register file (see RegFile): 256 in all = two waves per SIMD
#ifndef ZKW_MIN_WAVES_PER_SIMD
#define ZKW_MIN_WAVES_PER_SIMD 4
#endif
__global__ void __launch_bounds__(ZKW_WAVE * ZKW_MAX_WAVES_PER_GROUP, ZKW_MIN_WAVES_PER_SIMD) zkw_cycle_kernel(zkw_launch_args A) {
  // one wave = one independent group of L VM instances; the waves of a workgroup share the ISA table.  The waves of the
  // launch are numbered through all its batches (zkw_launch_args.wave_base): a workgroup may hold waves of two batches.
  const u32 tid = threadIdx.x % A.wave_threads;
  const u32 wib = zkw_uniform(threadIdx.x / A.wave_threads);  // a scalar: the per-wave bases below then live in SGPRs
  const u32 gw = blockIdx.x * A.waves_per_group + wib;         // wave of the launch
  u32 batch_idx, wave;
  if (A.uniform_waves) {
    batch_idx = gw / A.uniform_waves;
    wave = gw - batch_idx * A.uniform_waves;
  } else {  // largest b with wave_base[b] <= gw
    u32 lo = 0, hi = A.n_batches;
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (A.wave_base[mid] <= gw) lo = mid; else hi = mid;
    }
    batch_idx = lo;
    wave = gw - A.wave_base[lo];
  }
  const bool is_helper = A.helpers && wib >= A.waves_per_group;  // the extra wave(s) of the workgroup (zkw_dq_helper / zkw_kh_helper)
  const bool beyond = is_helper || gw >= A.wave_base[A.n_batches];  // tail of the last workgroup
  if (beyond) {
    batch_idx = 0;  // (a valid parameter block for the table staging below; the wave leaves after the barrier)
    wave = 0;
  }
  ZKW_KP P = *(const zkw_kparams ZKW_CONST_AS*)A.kp[batch_idx];
  Shared sh;
  shared_setup(sh, P, A.debug_flags, wib, wave, true);
#ifdef __HIP_DEVICE_COMPILE__
  zkw_set_lane_regs((u32)(size_t)(ZKW_LDS_AS u32*)sh.cold, (u32)(size_t)(ZKW_LDS_AS uint4*)sh.pcw, tid);  // v131..v133: see CF
#endif
  u32 run_cycles = A.run_cycles, time_delta = P.consts.time_delta_per_cycle, cap_delta = P.cap_delta, max_depth = P.consts.vm_max_stack_depth;
  ZKW_PIN_SGPR(run_cycles); ZKW_PIN_SGPR(time_delta); ZKW_PIN_SGPR(cap_delta); ZKW_PIN_SGPR(max_depth);
  u64 cond_lut = P.consts.condition_lut;  // Condition of the 3-bit field (cycle.rs:193-209): a table constant
  ZKW_PIN_SGPR(cond_lut);
  // stage the packed ISA table in LDS (all threads of the workgroup, 16 B each per step)
  {
    const uint4* src = (const uint4*)P.isa;
    uint4* dst = (uint4*)sh.isa;
    for (u32 i = threadIdx.x; i < ZKW_ISA_TABLE_SIZE / 2; i += blockDim.x) dst[i] = src[i];
  }
#ifdef ZKW_PROFILE
  for (u32 i = threadIdx.x; i < ZKW_MAX_WAVES_PER_GROUP * 80; i += blockDim.x) (&zp_acc[0][0])[i] = 0;
#endif
#ifdef ZKW_WAITPROF
  for (u32 i = threadIdx.x; i < ZKW_MAX_WAVES_PER_GROUP * 32; i += blockDim.x) (&zw_acc[0][0])[i] = 0;
#endif
#ifdef ZKW_WIDE
  if (is_helper) {  // counters and entries of every hand-over area start empty (before the barrier: the cycle waves post after it)
    u32* area = (u32*)(zkw_lds + ZKW_LDS_WAVES0 + A.waves_per_group * zkw_wave_lds_units());
    const u32 per_wave = (ZKW_DQ_HELPER_BYTES + ((A.debug_flags & ZKW_KECCAK_HELPER) ? ZKW_KH_BYTES : 0u)) / 4u;
    for (u32 i = tid + (wib - A.waves_per_group) * A.wave_threads; i < A.waves_per_group * per_wave; i += A.helpers * A.wave_threads) area[i] = 0;
  }
#endif
  __syncthreads();
#ifdef ZKW_WIDE
  if (is_helper) {
    if (A.debug_flags & ZKW_KECCAK_HELPER) {
      const u32 n_sub = A.helpers / A.waves_per_group, hx = wib - A.waves_per_group;
      zkw_kh_helper(A, tid, hx / n_sub, hx % n_sub, n_sub);
    }
    else zkw_dq_helper(A, tid);
    return;
  }
#endif
  if (beyond || wave >= P.n_waves) return;  // tail workgroup: no further workgroup-level barrier below
  if (tid == 0) {  // what zkw_heavy_entry needs and cannot take through its argument registers
    const u64 kp = (u64)A.kp[batch_idx];
    zkw_lds[ZKW_LDS_WAVES0 + wib * zkw_wave_lds_units() + 1] = make_uint4((u32)kp, (u32)(kp >> 32), A.debug_flags, wave);
#ifdef __HIP_DEVICE_COMPILE__
    const u32 area0 = (u32)(size_t)(ZKW_LDS_AS uint4*)(zkw_lds + ZKW_LDS_WAVES0 + A.waves_per_group * zkw_wave_lds_units());
    *ZKW_LDS_WORD((ZKW_LDS_AS u32*)sh.cursor) = A.helpers ? area0 + wib * ZKW_DQ_HELPER_BYTES : 0u;  // dq_helper_area
    *ZKW_LDS_WORD((ZKW_LDS_AS u32*)sh.cursor + 1) = area0 + A.waves_per_group * ZKW_DQ_HELPER_BYTES + wib * ZKW_KH_BYTES;  // kh_box (ZKW_KECCAK_HELPER)
    *ZKW_LDS_WORD((ZKW_LDS_AS u32*)sh.cursor + 2) = (u32)(size_t)(ZKW_LDS_AS char*)((char*)zkw_lds + A.lds_sink);  // the prefetch sink (op_uma)
#elif defined(ZKW_WIDE)
    const u32 area0 = ZKW_LDS_ADDR(zkw_lds + ZKW_LDS_WAVES0 + A.waves_per_group * zkw_wave_lds_units());
    *ZKW_LDS_WORD(sh.cursor) = A.helpers ? area0 + wib * ZKW_DQ_HELPER_BYTES : 0u;
    *ZKW_LDS_WORD(sh.cursor + 1) = area0 + A.waves_per_group * ZKW_DQ_HELPER_BYTES + wib * ZKW_KH_BYTES;
#endif
  }
  {  // the wave's stream cursors -> lanes 0..3 of v128 (see stream_alloc); uniform loads
    const u32* cp = P.cursors + wave * 4;
    zkw_cursor_set<0>(zkw_uniform(cp[0])); zkw_cursor_set<1>(zkw_uniform(cp[1])); zkw_cursor_set<2>(zkw_uniform(cp[2])); zkw_cursor_set<3>(zkw_uniform(cp[3]));
  }
#ifdef __HIP_DEVICE_COMPILE__
  const u32 lds_sink = zkw_lds_sink_addr();
#endif
  const u32 cycle_base = P.wave_cycles[wave];  // wave-cycles run since the reset (records / directory index)
  // first launch after a reset: the lanes start from the pristine images (the reset does not copy them — 2.5 MB per 4096
  // instances it would read and write); the write-back at the end of the launch fills the working buffers
  const bool fresh = cycle_base == 0;
  const zkw_dev_scalars* const scalars_in = fresh ? P.scalars0 : P.scalars;
  const uint4* const regs_in = fresh ? P.regs0 : P.regs;
  if (fresh && tid < P.L && wave * P.L + tid < P.n_instances) {
    // ... and the marks of what the previous run overwrote (the reset kernel has restored those words and slots; it only
    // reads the masks, so that it needs no ordering between its threads)
    const u32 groups = (P.heap_image_words + 31u) >> 5;
    for (u32 g = 0; g < groups; g++) P.heap_dirty[((u64)wave * groups + g) * P.L + tid] = 0;
    const u32 sw = (P.storage_slots + 31u) >> 5;
    for (u32 g = 0; g < sw; g++) P.storage_dirty[(u64)(wave * P.L + tid) * sw + g] = 0;
  }

  const u32 inst = wave * P.L + tid;
  const bool exists = tid < P.L && inst < P.n_instances;
  Lane s;
  s.lane = tid;
  s.pc = s.sp = s.ergs = s.timestamp = s.prev_super_pc = s.flags = s.kflags = s.ptr_bitmap = s.reg_dirty = s.counts = 0;
  RegFile rf;
  rf_init(rf);
#if defined(ZKW_EMU_BUILD) && defined(ZKW_WIDE)
  zkw_emu_rf_ptr[threadIdx.x] = &rf;
#endif
  if (exists) {
    const zkw_dev_scalars sc = scalars_in[inst];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint2 e = sh.isa[sc.prev_code_word[2 * i] & (ZKW_ISA_TABLE_SIZE - 1)];
      ZKW_SLOT_WRITE(sh, tid, i, make_uint4(sc.prev_code_word[2 * i], sc.prev_code_word[2 * i + 1], e.x, e.y));
    }
#pragma unroll
    for (int i = 0; i < 4; i++) CF(sh, s, CF_CTX0 + i) = sc.ctx_u128_reg[i];
    s.ptr_bitmap = sc.ptr_bitmap; s.flags = sc.flags; s.timestamp = sc.timestamp;
    CF(sh, s, CF_PREV_CODE_PAGE) = sc.prev_code_page; s.kflags = KF_CODE_PAGE_CHANGED;  // frame_load settles the marker
    CF(sh, s, CF_CYCLE_COUNTER0) = sc.cycle_counter; CF(sh, s, CF_SPENT_PUBDATA) = sc.spent_pubdata; CF(sh, s, CF_MPC) = sc.memory_page_counter; CF(sh, s, CF_ERGS_PP) = sc.ergs_per_pubdata;
    CF(sh, s, CF_TX_NUMBER) = sc.tx_number; s.prev_super_pc = sc.prev_super_pc; s.depth = sc.depth; s.status = sc.status; CF(sh, s, CF_N_CYCLES0) = sc.n_cycles;
    CF(sh, s, CF_FIRST_DYN) = sc.first_dynamic_page; CF(sh, s, CF_N_INITIAL_SLOTS) = sc.n_initial_slots; CF(sh, s, CF_NEXT_SLOT) = sc.next_slot; CF(sh, s, CF_JOURNAL_LEN) = sc.journal_len;
    CF(sh, s, CF_N_HISTORY) = sc.n_history;
    frame_load(P, sh, s);
    // register file -> v136..v255
    const uint4* rg = regs_in + (u64)wave * ZKW_REG_CHUNKS * P.L + tid;
    for (u32 r = 0; r < ZKW_REGISTERS_COUNT; r++) rf_set(rf, r + 1, u256_from_uint4(rg[(u64)(2 * r) * P.L], rg[(u64)(2 * r + 1) * P.L]));
  } else {
    s.status = ZKW_STATUS_ENDED;  // parked lane
    s.depth = 0;
  }

#ifdef ZKW_DEBUG_PRINT
  if (tid == 0 && gw == 0)
    printf("ZKWDBG exists %d depth %u status %u ergs %u pc %u code_page %u prev %u base %u code_len %u code_off %u kflags %x L %u wave %u wib %u inst %u\n", (int)exists, s.depth, s.status, s.ergs,
           s.pc, CF(sh, s, CF_CODE_PAGE), CF(sh, s, CF_PREV_CODE_PAGE), CF(sh, s, CF_BASE_PAGE), CF(sh, s, CF_CODE_LEN), CF(sh, s, CF_CODE_OFF), s.kflags, sh.L, sh.wave, sh.wib, inst);
#endif
  // running output pointers of this wave (advanced once per cycle instead of re-derived from the parameter block)
  u32* dir_ptr = P.dir + ((u64)wave * (P.max_cycles + 1) + cycle_base) * 4;
  uint4* const tails_wave = P.tails + ((u64)wave * P.max_cycles + cycle_base) * sh.L;  // wave-uniform
  uint4* const delta_base = P.deltas + (u64)wave * P.cap_delta * 2;
  const u32 tail_step = sh.L;
  // The cycle loop is entered once by the lanes that are running and left per lane (divergent exit) when the lane ends,
  // fails or has used its cycles: the lane state is then modified unconditionally inside the loop body instead of inside
  // an `if (active)` region of every iteration (whose merge points cost ~75 register copies per VM cycle).
  u32 k = 0;
  u32 delta_cur = zkw_cursor_get<3>();
  if (exists && !(s.status == ZKW_STATUS_RUNNING && run_cycles != 0 && s.depth != 0)) lane_writeback(P, sh, rf, s, 0);  // does not cycle
  ZKW_DIV_IF(exists && s.status == ZKW_STATUS_RUNNING && run_cycles != 0 && s.depth != 0) {  // (lanes leave the loop one by one and wait for the others behind this region)
    ZKW_PROF_DECL
#ifdef ZKW_SHORT_STATS
    // (scalars only: counters in an array would live in scratch memory, and every cycle would wait for its stores)
    u32 zs_short = 0, zs_odd = 0, zs_heavy = 0, zs_bad = 0, zs_cls = 0;
    u32 zs_n0 = 0, zs_n1 = 0, zs_n2 = 0, zs_n3 = 0, zs_n4 = 0, zs_n5 = 0;
    unsigned long long zs_t0 = 0, zs_t1 = 0, zs_t2 = 0, zs_t3 = 0, zs_t4 = 0, zs_t5 = 0, zs_last = __builtin_readcyclecounter();
    // class 0: general path, light opcode; 1: general path, UMA; 2: general path, log / near call / far call / ret; 3: short path, ALU; 4: short path, UMA; 5: short path, mul
#define ZKW_SS(x) x
#else
#define ZKW_SS(x)
#endif
    for (;;) {
      s.lane = zkw_lane_id();
#ifdef ZKW_SHORT_STATS
      if (k) {
        const unsigned long long zs_now = __builtin_readcyclecounter();
        const unsigned long long zs_d = zs_now - zs_last;
        if (zs_cls == 0) { zs_t0 += zs_d; zs_n0++; } else if (zs_cls == 1) { zs_t1 += zs_d; zs_n1++; } else if (zs_cls == 2) { zs_t2 += zs_d; zs_n2++; }
        else if (zs_cls == 3) { zs_t3 += zs_d; zs_n3++; } else if (zs_cls == 4) { zs_t4 += zs_d; zs_n4++; } else { zs_t5 += zs_d; zs_n5++; }
        zs_last = zs_now;
      }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZKW_SLEEP_PROBE) /* (experiment: N x 64 idle clocks per cycle — does the launch get longer by as much?) */
      __builtin_amdgcn_s_sleep(ZKW_SLEEP_PROBE);
#endif
#ifdef ZKW_ASM_MARKS
      asm volatile("; MARK loop top");
#endif
      ZKW_PROF_RESET
      // directory: stream cursors at the start of wave-cycle (cycle_base + k).  Read here (one broadcast 16-B LDS read),
      // stored by the first remaining lane after the fetch below, so that the LDS latency hides behind it.  The
      // register-delta cursor is carried in a register: only the end of the cycle advances it.
      const uint4 dir_entry = make_uint4(zkw_cursor_get<0>(), zkw_cursor_get<1>(), zkw_cursor_get<2>(), delta_cur);
      s.counts = 0; s.kflags &= ~(KF_COLD_DIRTY | KF_DQ_CHAINED); s.reg_dirty = 0;
      // ----------------------------------------------------------------------------------------
      // read_and_decode (cycle.rs:19-236)
      // ----------------------------------------------------------------------------------------
      const bool pending = (s.flags & FLAG_PENDING) != 0;
      const u32 super_pc = s.pc >> 2, sub_pc = s.pc & 3u;
      ZKW_DIV_IF(ZKW_LIKELY(!pending)) {
        ZKW_DIV_IF(ZKW_UNLIKELY((s.kflags & KF_CODE_PAGE_CHANGED) || s.prev_super_pc != super_pc)) {  // :59-95
          const u256 word = code_fetch(sh, s, super_pc);
          emit_mem(P, sh, s, s.timestamp, ZKW_MEM_CODE, CF(sh, s, CF_CODE_PAGE), super_pc, word, false, false, 0);
          // pre-decode the four opcodes of the word (four independent table reads) next to their encodings: a cycle
          // then needs ONE LDS read for its opcode and the packed ISA entry
          uint2 e4[4];
#pragma unroll
          for (int i = 0; i < 4; i++) e4[i] = sh.isa[word.w[2 * i] & (ZKW_ISA_TABLE_SIZE - 1)];
          ZKW_LGKM_PROBE(8 /* pre-decode table reads */)
#pragma unroll
          for (int i = 0; i < 4; i++) ZKW_SLOT_WRITE(sh, s.lane, i, make_uint4(word.w[2 * i], word.w[2 * i + 1], e4[i].x, e4[i].y));
          s.prev_super_pc = super_pc;
#ifdef __HIP_DEVICE_COMPILE__
          if (!ZKW_ABL(A.debug_flags, ZKW_NO_PREFETCH)) {  // this opcode and the ones behind it (opcode k of the word = slot 3 - k): prefetch_uma_words
#pragma unroll
            for (int i = 0; i < 4; i++)
              if (3u - (u32)i >= sub_pc) prefetch_uma_words(P, sh, s, lds_sink, e4[i].x, word.w[2 * i + 1]);
          }
#endif
        }
      } else {  // :104-115
        s.flags &= ~FLAG_PENDING;
        s.prev_super_pc = super_pc;
      }
#if !defined(ZKW_NO_FAST_ALU) /* (-DZKW_NO_FAST_ALU: the A/B partner) */
      // ------------------------------------------------------------------------------------------------------------
      // The short cycle.  A wave on a shared tape whose lanes all stand at the same pc, in kernel mode, with nothing pending,
      // executes here (a) an ALU instruction with register / immediate operands (nop, add, sub, and / or / xor, jump) and
      // (b) a heap / aux-heap access (uma.rs:26-425) that raises no exception and grows no bound: one slot read, scalar
      // decode, the operation, the record — no group loop, no operand addressing, no opcode switch, no out-of-line call site
      // on the path.  The fetch above is shared with the general path (a cycle that starts a code word qualifies like any
      // other).  Every test below is wave-uniform; a cycle that does not qualify (a memory operand, an exception, a growing
      // heap, a heavier opcode, diverged lanes) takes the general path underneath, untouched: nothing is written before the
      // cycle is known to qualify.  Same witness, bit for bit (cycle.rs:19-236 read_and_decode, :275-350 operands, add.rs /
      // sub.rs / binop.rs / jump.rs / noop.rs / uma.rs, :408-413).
      // ------------------------------------------------------------------------------------------------------------
#ifdef ZKW_ASM_MARKS
      asm volatile("; MARK short test");
#endif
      if (ZKW_LIKELY(k + 1u < run_cycles && !(A.debug_flags & (4u | (1u << 24))))) {  // (the last cycle of a launch leaves through the general path; test hooks: general path)
        const u32 pc0 = (u32)__builtin_amdgcn_readfirstlane((int)s.pc);
        const bool odd = (s.pc != pc0) | pending | ((s.kflags & (KF_TAIL2 | KF_STATIC | KF_KERNEL)) != KF_KERNEL) | (s.depth == max_depth) | !lane_ok(s);
        if (zkw_ballot(odd) == 0) {
          const uint4 me = ZKW_SLOT_READ(sh, s.lane, 3u - (pc0 & 3u));
          const u32 u_lo = (u32)__builtin_amdgcn_readfirstlane((int)me.x), u_hi = (u32)__builtin_amdgcn_readfirstlane((int)me.y);
          const u32 u_attr = (u32)__builtin_amdgcn_readfirstlane((int)me.z), u_price = (u32)__builtin_amdgcn_readfirstlane((int)me.w);
          const u32 opcode = ZKW_ATTR_OPCODE(u_attr), props = ZKW_ATTR_PROPS(u_attr), src0_mode = ZKW_ATTR_SRC0(u_attr), variant = ZKW_ATTR_VARIANT(u_attr);
#ifdef ZKW_SHORT_CLASS /* (-DZKW_SHORT_CLASS, the A/B partner of round 6: the class of the instruction from the bits the host packed into its
                          ISA entry — zkw_short_class, zkw_device.h — instead of 22 compare / select pairs per cycle: profiles/r10_short_cycle_census.txt) */
          const bool alu = (u_attr & ZKW_ATTR_SHORT_ALU) != 0;
          const bool two_regs = opcode == ZKW_OP_MUL;  // (dst0 and dst1)
          const bool code_operand = (u_attr & ZKW_ATTR_SHORT_CODE) != 0;
#ifndef ZKW_NO_FAST_UMA
          const bool uma = (u_attr & ZKW_ATTR_SHORT_UMA) != 0;
#ifdef ZKW_SHORT_STACK /* (-DZKW_SHORT_STACK, A/B partner: ALU instructions with stack operands — mem_ops.rs:51-121 — run in the short cycle too) */
          const bool stack_op = (u_attr & ZKW_ATTR_SHORT_STACK) != 0;
          const bool light = (u_attr & (ZKW_ATTR_SHORT_OK | ZKW_ATTR_SHORT_STACK)) && delta_cur + 2u * ZKW_WAVE <= cap_delta &&
                             (!(u_attr & (ZKW_ATTR_SHORT_MEM | ZKW_ATTR_SHORT_STACK)) || zkw_cursor_get<0>() + 4u * ZKW_WAVE <= sh.cap_mem);
#else
          const bool light = (u_attr & ZKW_ATTR_SHORT_OK) && delta_cur + 2u * ZKW_WAVE <= cap_delta && (!(u_attr & ZKW_ATTR_SHORT_MEM) || zkw_cursor_get<0>() + 4u * ZKW_WAVE <= sh.cap_mem);
#endif
#else
          const bool uma = false;
          const bool light = (u_attr & ZKW_ATTR_SHORT_OK) && !(u_attr & ZKW_ATTR_SHORT_UMA) && delta_cur + 2u * ZKW_WAVE <= cap_delta &&
                             (!code_operand || zkw_cursor_get<0>() + 4u * ZKW_WAVE <= sh.cap_mem);
#endif
          (void)alu;
#else
          // (div stays with the general path: a second inlined u256_divmod in the loop made the driver's command 0.7 % slower, profiles/r08_ab_log.txt)
          const bool alu = ((1u << opcode) & ((1u << ZKW_OP_NOP) | (1u << ZKW_OP_ADD) | (1u << ZKW_OP_SUB) | (1u << ZKW_OP_MUL) | (1u << ZKW_OP_JUMP) | (1u << ZKW_OP_SHIFT) |
                                              (1u << ZKW_OP_BINOP))) != 0;
          const bool two_regs = opcode == ZKW_OP_MUL;  // (dst0 and dst1)
          const bool code_operand = src0_mode == ZKW_MODE_CODE && alu && opcode != ZKW_OP_NOP;  // a constant from the code page (mem_ops.rs:100-110)
#ifndef ZKW_NO_FAST_UMA
          const bool uma = opcode == ZKW_OP_UMA && variant <= ZKW_UMA_AUX_WRITE && !(props & ZKW_PROP_SWAP);
#else
          const bool uma = false;
#endif
          const bool light = (alu || uma) && ZKW_ATTR_DST0(u_attr) == ZKW_MODE_REG && (src0_mode == ZKW_MODE_REG || src0_mode == ZKW_MODE_IMM || code_operand) &&
                             !(props & ZKW_PROP_EXPLICIT_PANIC) && delta_cur + 2u * ZKW_WAVE <= cap_delta &&
                             (!(uma || code_operand) || zkw_cursor_get<0>() + 4u * ZKW_WAVE <= sh.cap_mem);  // (its four queries at most cannot run out of stream)
#endif
          if (light) {
            const u32 r_src0 = (u_lo >> 16) & 15u, r_src1 = (u_lo >> 20) & 15u, r_dst0 = (u_lo >> 24) & 15u, r_dst1 = u_lo >> 28;
            const bool run = condition_resolved(cond_lut, (u_lo >> 13) & 7u, s.flags);          // :193-217: a lane whose condition fails runs a nop
            // (the ISA entry follows from the opcode word: one table per batch, one batch per wave)
            bool bad = (me.x != u_lo) | (me.y != u_hi) | (s.ergs < u_price);
            const bool uma_heap = variant == ZKW_UMA_HEAP_READ || variant == ZKW_UMA_HEAP_WRITE;
            const bool uma_write = variant == ZKW_UMA_HEAP_WRITE || variant == ZKW_UMA_AUX_WRITE;
            u256 a = u256_zero();
#if defined(ZKW_SHORT_CLASS) && defined(ZKW_SHORT_STACK)
            // stack operands (MemOpsProcessor::compute_addresses_and_select_operands, mem_ops.rs:14-125): src0's location first, then dst0's, each
            // from its register + immediate; push / pop move sp.  A NOP moves sp and touches no memory (cycle.rs:298-301).
            const u32 dst0_mode = ZKW_ATTR_DST0(u_attr);
            const bool src0_stack = stack_op && src0_mode != ZKW_MODE_REG && src0_mode != ZKW_MODE_IMM, dst0_stack = stack_op && dst0_mode != ZKW_MODE_REG;
            u32 sp_new = s.sp, idx0 = 0, idx1 = 0;
            if (opcode != ZKW_OP_NOP || stack_op) a = src0_mode == ZKW_MODE_IMM ? u256_from_u32(u_hi & 0xffffu) : rf_get(rf, r_src0);
            if (src0_stack) {
              const u32 vaddr = (clip16(sh, a) + (u_hi & 0xffffu)) & 0xffffu;
              if (src0_mode == ZKW_MODE_STACK_PP) { sp_new = (sp_new - vaddr) & 0xffffu; idx0 = sp_new; }
              else if (src0_mode == ZKW_MODE_STACK_OFF) idx0 = (sp_new - vaddr) & 0xffffu;
              else idx0 = vaddr;
            }
            if (dst0_stack) {
              const u32 vaddr = (clip16(sh, rf_get(rf, r_dst0)) + (u_hi >> 16)) & 0xffffu;
              if (dst0_mode == ZKW_MODE_STACK_PP) { idx1 = sp_new; sp_new = (sp_new + vaddr) & 0xffffu; }
              else if (dst0_mode == ZKW_MODE_STACK_OFF) idx1 = (sp_new - vaddr) & 0xffffu;
              else idx1 = vaddr;
              bad |= run & (opcode != ZKW_OP_NOP) & (idx1 >= sh.S);  // (a write beyond the stack's capacity is a limit status: the general path reports it)
            }
#else
            if (opcode != ZKW_OP_NOP) a = src0_mode == ZKW_MODE_IMM ? u256_from_u32(u_hi & 0xffffu) : rf_get(rf, r_src0);  // (a code operand: the register part of its address)
#endif
            if (uma) {  // uma.rs:121-207: the offset dereferenceable, no overflow of the increment, inside the bound paid for, inside the arena
              const u32 off = a.w[0], inc = off + 32u;
              const u32 bound = uma_heap ? cfv_heap_bound(sh, s) : cfv_aux_bound(sh, s);
              bad |= run & (((a.w[1] | a.w[2] | a.w[3] | a.w[4] | a.w[5] | a.w[6] | a.w[7]) != 0) | (off > sh.max_deref_low) | (inc < off) | (inc >= bound) |
                            ((off >> 5) + 1u >= (uma_heap ? sh.H : sh.A)));
            }
            if (zkw_ballot(bad) == 0) {
#ifdef ZKW_ASM_MARKS
      asm volatile("; MARK short qualified");
#endif
              // ---- the cycle qualifies: from here on it is executed here ----
              ZKW_SS(zs_short++; zs_cls = uma ? 4u : (two_regs ? 5u : 3u);)
              ZKW_EMU_COUNT(0);
              if (uma) ZKW_EMU_COUNT(1);
#ifndef ZKW_EXP_NODIR /* (experiment: what the directory store costs — wrong results) */
              if (zkw_rank_below(zkw_ballot(1)) == 0) *(uint4*)dir_ptr = dir_entry;
#endif
              s.kflags = (s.kflags & ~(KF_CODE_PAGE_CHANGED | KF_MASKED)) | KF_CHARGED;
              s.ergs -= u_price;                                                                  // :153-161
              u32 new_pc = (s.pc + 1u) & 0xffffu;
              u32 dm = 0;
              u256 res = u256_zero();
              if (uma) {
                const u32 increment = ZKW_ATTR_FLAGS(u_attr) & 1u;
                ZKW_DIV_IF(run) {
                  const u32 f_slot = cfv_slot(sh, s);
                  const u32 f_hwm_in = uma_heap ? cfv_heap_hwm(sh, s) : CF(sh, s, CF_AUX_HWM);
                  u32 f_hwm = f_hwm_in;
                  const u32 page = CF(sh, s, CF_BASE_PAGE) + (uma_heap ? 2u : 3u);
                  const u32 mem_type = uma_heap ? ZKW_MEM_HEAP : ZKW_MEM_AUX_HEAP;
                  const u32 off = a.w[0], word0 = off >> 5, unal = off & 31u;
                  const u32 src0_ptr = src0_mode == ZKW_MODE_REG ? ((s.ptr_bitmap << 1) >> r_src0) & 1u : 0u;
                  const u32 ts_r = s.timestamp, ts_w = s.timestamp + 3u;
#ifdef ZKW_EXP_NOLOAD /* (experiment: what the exposed latency of the word loads costs — wrong results) */
                  u256 w0v = u256_from_u32(word0), w1v = u256_zero();
#else
                  u256 w0v = heap_read_at(P, sh, s, !uma_heap, f_slot, f_hwm, word0), w1v = u256_zero();
                  if (unal) w1v = heap_read_at(P, sh, s, !uma_heap, f_slot, f_hwm, word0 + 1u);
#endif
                  ZKW_SETTLE(2 /* UMA words */);
#ifdef ZKW_ASM_MARKS
                  asm volatile("; MARK emit0 begin");
#endif
#ifndef ZKW_EXP_NOEMIT /* (experiment: what the read queries cost — wrong results) */
                  emit_mem(P, sh, s, ts_r, mem_type, page, word0, w0v, false, false, 0);
                  ZKW_DIV_IF(unal) emit_mem(P, sh, s, ts_r, mem_type, page, word0 + 1u, w1v, false, false, 0);
#endif
#ifdef ZKW_ASM_MARKS
                  asm volatile("; MARK emit0 end");
#endif
                  const bool all_aligned = zkw_ballot(unal != 0) == 0;
                  const u32 u_unal = (u32)__builtin_amdgcn_readfirstlane((int)unal);
                  const bool same_unal = zkw_ballot(unal != u_unal) == 0;
                  const u32 u_b8 = (u_unal & 3u) * 8u;
                  u256 upd = u256_zero();
                  upd.w[0] = off + 32u;
                  if (!uma_write) {  // uma.rs:291-348
                    if (all_aligned) {
                      res = w0v;
                    } else if (same_unal) {
                      switch (u_unal >> 2) {
                        case 0: res = u256_byte_window_at<0>(w0v, w1v, u_b8); break;
                        case 1: res = u256_byte_window_at<1>(w0v, w1v, u_b8); break;
                        case 2: res = u256_byte_window_at<2>(w0v, w1v, u_b8); break;
                        case 3: res = u256_byte_window_at<3>(w0v, w1v, u_b8); break;
                        case 4: res = u256_byte_window_at<4>(w0v, w1v, u_b8); break;
                        case 5: res = u256_byte_window_at<5>(w0v, w1v, u_b8); break;
                        case 6: res = u256_byte_window_at<6>(w0v, w1v, u_b8); break;
                        default: res = u256_byte_window_at<7>(w0v, w1v, u_b8); break;
                      }
                    } else {
                      res = u256_byte_window(w0v, w1v, unal);
                    }
                    if (r_dst0 != 0) {
                      rf_set(rf, r_dst0, res);
                      dm = 1u << (r_dst0 - 1u);
                      s.ptr_bitmap &= ~dm;
                    }
                    if (increment && r_dst1 != 0) {  // (l[0] & TOP_32) + incremented :337-338; the pointer tag of src0 travels with it
                      rf_set(rf, r_dst1, upd);
                      dm |= 1u << (r_dst1 - 1u);
                      s.ptr_bitmap = (s.ptr_bitmap & ~(1u << (r_dst1 - 1u))) | (src0_ptr << (r_dst1 - 1u));
                    }
                  } else {  // uma.rs:349-423
                    const u256 b = rf_get(rf, r_src1);
                    u256 n0, n1;
                    if (all_aligned) {
                      n0 = b;
                      n1 = u256_zero();
                    } else if (same_unal) {
                      switch (u_unal >> 2) {
                        case 0: u256_merge_at<0>(w0v, w1v, b, u_b8, n0, n1); break;
                        case 1: u256_merge_at<1>(w0v, w1v, b, u_b8, n0, n1); break;
                        case 2: u256_merge_at<2>(w0v, w1v, b, u_b8, n0, n1); break;
                        case 3: u256_merge_at<3>(w0v, w1v, b, u_b8, n0, n1); break;
                        case 4: u256_merge_at<4>(w0v, w1v, b, u_b8, n0, n1); break;
                        case 5: u256_merge_at<5>(w0v, w1v, b, u_b8, n0, n1); break;
                        case 6: u256_merge_at<6>(w0v, w1v, b, u_b8, n0, n1); break;
                        default: u256_merge_at<7>(w0v, w1v, b, u_b8, n0, n1); break;
                      }
                    } else {
                      const u32 lowest = 32 - unal;
                      n0 = u256_shl(u256_shr(w0v, lowest * 8), lowest * 8);
                      n0 = u256_or(n0, u256_shr(b, unal * 8));
                      n1 = u256_shr(u256_shl(w1v, unal * 8), unal * 8);
                      n1 = u256_or(n1, u256_shl(b, (32 - unal) * 8));
                    }
                    heap_write_at(P, sh, s, !uma_heap, f_slot, f_hwm, word0, n0);
                    emit_mem(P, sh, s, ts_w, mem_type, page, word0, n0, false, true, 0);
                    ZKW_DIV_IF(unal) {
                      heap_write_at(P, sh, s, !uma_heap, f_slot, f_hwm, word0 + 1u, n1);
                      emit_mem(P, sh, s, ts_w, mem_type, page, word0 + 1u, n1, false, true, 0);
                    }
                    if (ZKW_UNLIKELY(f_hwm != f_hwm_in)) {
                      if (uma_heap) cfv_set_heap_hwm(sh, s, f_hwm); else CF(sh, s, CF_AUX_HWM) = f_hwm;
                    }
                    if (increment && r_dst0 != 0) {
                      rf_set(rf, r_dst0, upd);
                      dm = 1u << (r_dst0 - 1u);
                      s.ptr_bitmap &= ~dm;
                    }
                  }
#ifdef __HIP_DEVICE_COMPILE__
                  // (a cursor: the next access through the register is 32 bytes on — see op_uma)
                  if (increment && src0_mode != ZKW_MODE_IMM && !ZKW_ABL(sh.debug_flags, ZKW_NO_PREFETCH))
                    prefetch_page_words(uma_heap ? sh.heap : sh.aux_heap, uma_heap ? P.H : P.A, P.L, zkw_lds_sink_addr(), f_slot, f_hwm, off + 32u);
#endif
                }
              } else if (opcode != ZKW_OP_NOP) {
                if (code_operand) {  // cycle.rs:304-325: the word of the code page at (register + imm0), and its query
                  ZKW_DIV_IF(run) {
                    const u32 idx = (clip16(sh, a) + (u_hi & 0xffffu)) & 0xffffu;  // mem_ops.rs:34-35
                    a = code_fetch(sh, s, idx);
                    emit_mem(P, sh, s, s.timestamp, ZKW_MEM_CODE, CF(sh, s, CF_CODE_PAGE), idx, a, false, false, 0);
                  }
                }
#if defined(ZKW_SHORT_CLASS) && defined(ZKW_SHORT_STACK)
                if (src0_stack) {  // cycle.rs:304-325: the operand from the stack of the current frame, and its query
                  ZKW_DIV_IF(run) {
                    u32 tag;
                    a = stack_read(P, sh, s, idx0, tag);
                    emit_mem(P, sh, s, s.timestamp, ZKW_MEM_STACK, CF(sh, s, CF_BASE_PAGE) + 1u, idx0, a, tag != 0, false, 0);
                  }
                }
#endif
                u256 b = rf_get(rf, r_src1);
                if (props & ZKW_PROP_SWAP) {  // :341-345 (wave-uniform)
                  const u256 t = a;
                  a = b;
                  b = t;
                }
                if (opcode == ZKW_OP_JUMP) {
                  if (run) new_pc = clip16(sh, a);  // jump.rs:23-25
                } else {
                  bool of = false, eq = false, gt = false;
                  u256 res1 = u256_zero();  // dst1 of mul
                  if (opcode == ZKW_OP_ADD) {  // add.rs:35-53
                    res = u256_add(a, b, of);
                    eq = u256_is_zero(res); gt = !eq && !of;
                  } else if (opcode == ZKW_OP_SUB) {  // sub.rs:35-54
                    res = u256_sub(a, b, of);
                    eq = u256_is_zero(res); gt = !eq && !of;
                  } else if (opcode == ZKW_OP_BINOP) {  // binop.rs:42-61
                    res = variant == ZKW_BINOP_XOR ? u256_xor(a, b) : (variant == ZKW_BINOP_AND ? u256_and(a, b) : u256_or(a, b));
                    eq = u256_is_zero(res);
                  } else if (opcode == ZKW_OP_SHIFT) {  // shift.rs:44-78
                    const u32 n = b.w[0] & 0xffu;
                    const bool cyclic = variant == ZKW_SHIFT_ROL || variant == ZKW_SHIFT_ROR, right = variant == ZKW_SHIFT_SHR || variant == ZKW_SHIFT_ROR;
                    if (right) {
                      res = u256_shr(a, n);
                      if (cyclic) res = u256_or(res, u256_shl(a, 256u - n));
                    } else {
                      res = u256_shl(a, n);
                      if (cyclic) res = u256_or(res, u256_shr(a, 256u - n));
                    }
                    eq = u256_is_zero(res);
                  } else {  // mul.rs:35-65
                    if (run) u256_mul(a, b, res, res1);
                    of = !u256_is_zero(res1); eq = u256_is_zero(res); gt = !of && !eq;
                  }
#if defined(ZKW_SHORT_CLASS) && defined(ZKW_SHORT_STACK)
                  if (dst0_stack) {  // perform_dst0_update, helpers.rs:266-283: the result goes to the stack, with its query
                    ZKW_DIV_IF(run) {
                      stack_write(P, sh, s, idx1, res, false);
                      emit_mem(P, sh, s, s.timestamp + 3u, ZKW_MEM_STACK, CF(sh, s, CF_BASE_PAGE) + 1u, idx1, res, false, true, 0);
                    }
                  }
#endif
                  if (run) {
                    if (ZKW_ATTR_FLAGS(u_attr) & 1u) set_flags3(s, of, eq, gt);
#if defined(ZKW_SHORT_CLASS) && defined(ZKW_SHORT_STACK)
                    if (r_dst0 != 0 && !dst0_stack) {
#else
                    if (r_dst0 != 0) {
#endif
                      rf_set(rf, r_dst0, res);
                      dm = 1u << (r_dst0 - 1u);
                    }
                    if (two_regs && r_dst1 != 0) {
                      rf_set(rf, r_dst1, res1);
                      dm |= 1u << (r_dst1 - 1u);
                    }
                    s.ptr_bitmap &= ~dm;
                  }
                }
              }
#ifdef ZKW_ASM_MARKS
      asm volatile("; MARK short record");
#endif
              s.lane = zkw_lane_id();
              s.pc = new_pc;
#if defined(ZKW_SHORT_CLASS) && defined(ZKW_SHORT_STACK)
              if (stack_op && run) s.sp = sp_new;  // cycle.rs:297
#endif
#ifdef __HIP_DEVICE_COMPILE__
              asm("v_add_u32 %0, %1, %0" : "+v"(s.timestamp) : "s"(time_delta));  // :408-411
#else
              s.timestamp += time_delta;
#endif
              // CycleRecord: the registers this cycle wrote (ascending, lanes in lane order within a register: every lane
              // that wrote holds the same mask), then the tails
              const u64 part = zkw_ballot(dm != 0);
              const u32 per_reg = (u32)__popcll(part);
              u32 pos = delta_cur;
              if (per_reg) {
                const u32 any = (u32)__builtin_amdgcn_readlane((int)dm, (int)((u32)__ffsll((long long)part) - 1u));
                const u32 my = pos + zkw_rank_below(part);
                for (u32 left = any; left; left &= left - 1u) {
                  const u32 r = (u32)__ffsll((long long)left) - 1u;
                  if (dm) {
                    const u256 v = (uma || two_regs) ? rf_get(rf, r + 1u) : res;  // (wave-uniform choice: a one-register ALU cycle still holds its value)
                    const u32 at = my + (pos - delta_cur);
                    zkw_stream_store(delta_base + (u64)at, u256_lo4(v));
                    zkw_stream_store(delta_base + (u64)cap_delta + at, u256_hi4(v));
                  }
                  pos += per_reg;
                }
              }
#ifndef ZKW_EXP_NOTAIL /* (experiment: what the tail store costs — wrong results) */
              zkw_stream_store(tails_wave + (u64)k * tail_step + s.lane,
                               make_uint4((s.ptr_bitmap & 0xffffu) | ((s.flags & 0xfu) << 16) | ((dm & 0xffu) << 24), (s.pc & 0xffffu) | (s.sp << 16), s.ergs,
                                          (s.counts >> 8) | ((dm >> 8) << 24)));
#endif
              if (pos != delta_cur) {
                delta_cur = pos;
                zkw_cursor_set<3>(delta_cur);
              }
#ifdef ZKW_ASM_MARKS
      asm volatile("; MARK short end");
#endif
              k++;
              dir_ptr += 4;
              continue;
            }
            ZKW_SS(else zs_bad++;)
          }
          ZKW_SS(else zs_heavy++;)
        }
        ZKW_SS(else zs_odd++;)
      }
#endif
      s.kflags &= ~(KF_CODE_PAGE_CHANGED | KF_CHARGED | KF_MASKED);  // previous_code_memory_page := code_page (:49)
      if (ZKW_UNLIKELY(pending)) {  // the instruction is exception_revert_encoding() instead of the slot of the code word (:104-115)
        const u64 rv = P.consts.exception_revert_encoding;
        const uint2 e0 = sh.isa[(u32)rv & (ZKW_ISA_TABLE_SIZE - 1)];
        ZKW_SLOT_WRITE(sh, s.lane, 4, make_uint4((u32)rv, (u32)(rv >> 32), e0.x, e0.y));
        s.kflags |= KF_MASKED;
      }
      {
        const u64 in_loop = zkw_ballot(1);
        if (zkw_rank_below(in_loop) == 0) *(uint4*)dir_ptr = dir_entry;
      }
      // ----------------------------------------------------------------------------------------
      // decode + execute, grouped by instruction word: take the first lane that has
      // not been served, broadcast its opcode word (readlane -> SGPRs), ballot the lanes holding the
      // same word, decode ONCE on the scalar unit and run the body with exec = that group.  A shared
      // tape needs one iteration per cycle; lanes running different programs need one per distinct word.
      // Lanes whose decode raises an exception (masked into panic, cycle.rs:187-190) or whose condition
      // fails (masked into nop, :212-217) are served by extra passes with the panic / nop variant.
      // The instruction of a lane — opcode word + packed ISA entry, 16 bytes — is read from LDS at the top of every
      // iteration (its pre-decoded slot of the code word, integer_representaiton_from_u256: opcode k of a word is u64
      // limb 3-k, :86-94 — or sh.enc once the lane was masked): held in registers across the opcode bodies these four
      // values are what the allocator spills to scratch memory.
      // ----------------------------------------------------------------------------------------
      ZKW_PROF(0)  // fetch, directory
      u64 todo = zkw_ballot(1);
      while (todo) {
        const u32 leader = (u32)__ffsll((long long)todo) - 1u;
        const u32 lane_now = zkw_lane_id();
        // (a lane that was already served reads a slot it no longer cares about: it is not in `todo`)
        const uint4 me = ZKW_SLOT_READ(sh, lane_now, (s.kflags & KF_MASKED) ? 4u : 3u - (s.pc & 3u));
        ZKW_LGKM_PROBE(6 /* instruction slot */)
        const u32 charged = s.kflags & KF_CHARGED;
        const u32 u_lo = (u32)__builtin_amdgcn_readlane((int)me.x, (int)leader);
        const u32 u_hi = (u32)__builtin_amdgcn_readlane((int)me.y, (int)leader);
        const u32 u_charged = (u32)__builtin_amdgcn_readlane((int)charged, (int)leader);
        // only lanes that are still waiting: a lane that already ran a genuine `nop` must not join the group of lanes
        // that were masked into the nop encoding later in the same cycle.  The group is kept as a wave MASK (a scalar
        // register pair: compares write masks, masks combine on the scalar unit, a mask is a branch condition as it is) —
        // as a per-lane bool every step of this selection went through a v_cndmask / v_cmp round trip
        // (one ballot per compare, combined on the scalar unit: the ballot of an `&&` goes through a per-lane 0 / 1 again)
        const u64 same_state = todo & zkw_ballot(charged == u_charged);
        u64 grp = same_state & zkw_ballot(me.x == u_lo) & zkw_ballot(me.y == u_hi);
        const u32 u_attr = (u32)__builtin_amdgcn_readlane((int)me.z, (int)leader);
        const u32 u_price = (u32)__builtin_amdgcn_readlane((int)me.w, (int)leader);
        // Variant grouping.  When lanes are left over that do not hold the leader's word (the wave runs different
        // programs), widen the group to every waiting lane with the leader's ISA entry — same opcode, variant, addressing
        // modes and price, any register numbers / immediates / condition — if that serves more lanes: their cycle then
        // runs with per-lane operand decode (zkw_vec_exec).  Not for the heavy opcodes, whose out-of-line bodies take the
        // instruction word as a scalar.  A shared tape never gets here (its first group is all of `todo`).
        bool vec = false;
        if (ZKW_UNLIKELY(A.debug_flags & 4u)) {
          grp = 1ull << leader;  // test hook: one lane per group
        }
#ifdef ZKW_WIDE
        else if (ZKW_UNLIKELY(grp != todo || (A.debug_flags & (1u << 24)))) {  // (bit 24: every group the variant way — test hook)
          const u32 u_op = ZKW_ATTR_OPCODE(u_attr);
          if (u_op != ZKW_OP_LOG && u_op != ZKW_OP_NEAR_CALL && u_op != ZKW_OP_FAR_CALL && u_op != ZKW_OP_RET &&
              !((A.debug_flags >> (8u + u_op)) & 1u)) {  // (debug_flags bits 8..23: opcodes kept out of variant groups — test hook)
            const u64 wide = same_state & zkw_ballot(me.z == u_attr) & zkw_ballot(me.w == u_price);
            if (__popcll(wide) > __popcll(grp) || (A.debug_flags & (1u << 24))) {
              grp = wide;
              vec = true;
            }
          }
        }
#endif
        if (ZKW_LIKELY(!u_charged)) {  // uniform: first visit of this opcode word
          bool masked_now = false;
          if (zkw_lane_bit(grp)) {
            const bool err = decode_exception(max_depth, s, u_attr, u_price);  // :142-184
            if (s.ergs < u_price) s.ergs = 0; else s.ergs -= u_price;  // :153-161
            const bool nop = !err && !condition_resolved(cond_lut, (me.x >> 13) & 7u, s.flags);  // (the lane's own condition: a variant group mixes them)
            s.kflags |= KF_CHARGED;
            masked_now = err | nop;
            if (ZKW_UNLIKELY(masked_now)) {
              // mask_into_panic (:187-190) / mask_into_nop (:212-217): the lane re-enters the loop as a member of
              // the group of the panic / nop encoding (all operand fields zero, condition Always)
              // (both encodings as scalar loads, then a per-lane select: `err ? a : b` on the parameter block itself became a
              // per-lane ADDRESS select and a vector-memory load + s_waitcnt vmcnt(0) for every masked group)
              const u64 enc_panic = P.consts.exception_revert_encoding, enc_nop = P.consts.nop_encoding;
              const u64 masked = err ? enc_panic : enc_nop;
              const uint2 e1 = sh.isa[(u32)masked & (ZKW_ISA_TABLE_SIZE - 1)];
              ZKW_SLOT_WRITE(sh, lane_now, 4, make_uint4((u32)masked, (u32)(masked >> 32), e1.x, e1.y));
              s.kflags |= KF_MASKED;
            }
          }
          grp &= ~zkw_ballot(masked_now);  // (they stay in `todo`)
        }
        todo &= ~grp;
        const bool mine = zkw_lane_bit(grp);
        ZKW_PROF(1)  // group selection, price, exceptions, condition
        ZKW_DIV_IF(mine) {
          ZKW_EMU_COUNT(2);
          if (vec) ZKW_EMU_COUNT(3);
          Decoded d;
          d.word_lo = u_lo; d.word_hi = u_hi;
          d.attr = u_attr;
          d.cond = (u_lo >> 13) & 7u; d.src0 = (u_lo >> 16) & 15u; d.src1 = (u_lo >> 20) & 15u; d.dst0 = (u_lo >> 24) & 15u; d.dst1 = u_lo >> 28;
          d.imm0 = u_hi & 0xffffu; d.imm1 = u_hi >> 16;
          if (ZKW_ABL(A.debug_flags, 8u)) s.pc = (s.pc + 1u) & 0xffffu;  // profiling ablation: no operand / opcode work
          else exec_decoded(P, sh, rf, s, d, vec, me.x, me.y);
          ZKW_SS({ const u32 zo = ZKW_ATTR_OPCODE(u_attr); zs_cls = zo == ZKW_OP_UMA ? 1u : ((zo == ZKW_OP_LOG || zo == ZKW_OP_NEAR_CALL || zo == ZKW_OP_FAR_CALL || zo == ZKW_OP_RET) ? 2u : 0u); })
#ifdef ZKW_PROFILE
          {
            const unsigned long long zp_now = __builtin_readcyclecounter();
            if (zkw_rank_below(zkw_ballot(1)) == 0) {
              zp_acc[sh.wib][8 + (ZKW_ATTR_OPCODE(u_attr) & 15u)] += zp_now - zp_last;
              zp_acc[sh.wib][24 + (ZKW_ATTR_OPCODE(u_attr) & 15u)] += 1;
            }
            zp_last = zp_now;
          }
#endif
        }
      }
      // ----------------------------------------------------------------------------------------
      // end of cycle (cycle.rs:408-413)
      // ----------------------------------------------------------------------------------------
      ZKW_DIV_IF(lane_ok(s)) {
#ifdef __HIP_DEVICE_COMPILE__
        // the scalar operand spelled out: left to itself the optimiser hoists a vector copy of `time_delta` out of the
        // loop and then keeps that copy in scratch memory
        asm("v_add_u32 %0, %1, %0" : "+v"(s.timestamp) : "s"(time_delta));
#else
        s.timestamp += time_delta;
#endif
        ZKW_DIV_IF(ZKW_UNLIKELY(s.kflags & KF_COLD_DIRTY)) {
          uint4* a = aux_alloc(P, sh, s, ZKW_AUX_COLD_STATE, 0, CF(sh, s, CF_SPENT_PUBDATA), CF(sh, s, CF_ERGS_PP), CF(sh, s, CF_TX_NUMBER));
          if (a) {
            a[1] = make_uint4(CF(sh, s, CF_CTX0 + 0), CF(sh, s, CF_CTX0 + 1), CF(sh, s, CF_CTX0 + 2), CF(sh, s, CF_CTX0 + 3));
            a[2] = make_uint4(CF(sh, s, CF_MPC), 0, 0, 0);
          }
        }
      }
      if (!ZKW_ABL(A.debug_flags, 1u)) {
        // CycleRecord, delta form: the 512-byte snapshot the tracer observes (15 registers + 32-byte tail) is emitted as
        // the tail (dense [cycle][lane], coalesced) plus the 32-byte values of the registers THIS cycle wrote, compacted
        // per wave.  The delta of register r of a lane sits at base + (deltas of registers below r in this wave-cycle) +
        // (rank of the lane among the lanes that wrote r): no tags and no atomics — the host (and any consumer)
        // recomputes the positions from the dirty masks in the tails and rebuilds the snapshots from the initial
        // register file.  A cycle writes one register on average, so this is ~70 B instead of 512 B.
        // Order inside a wave-cycle: by register (ascending), lanes in lane order within a register — the register
        // index of a store is then wave-uniform (the values come straight from the VGPR register file).
        ZKW_PROF(2)  // end-of-cycle bookkeeping
        s.lane = zkw_lane_id();
        const bool ok = lane_ok(s);
        // (bit 15 of the mask: the "register" that carries heap bound, aux-heap bound and callstack depth — the half of the
        // record tail that changes a few times per hundred cycles travels as a delta like a register, 32 B when it changes
        // instead of 16 B in every tail; timestamp and previous_super_pc are not stored at all: the one advances by a
        // constant per cycle, the other is the pc the cycle started from — the host rebuilds both, zkw_runtime.cpp)
        const u32 dm = ok ? (s.reg_dirty | ((s.kflags & KF_TAIL2) ? 0x8000u : 0u)) : 0u;
        // union of the lanes' dirty masks and the number of deltas of this wave-cycle; a shared tape makes all masks equal
        const u32 dm0 = (u32)__builtin_amdgcn_readfirstlane((int)dm);
        u32 any, total;
        if (ZKW_LIKELY(zkw_ballot(dm != dm0) == 0)) {
          any = dm0;
          total = (u32)__popcll((u64)dm0) * (u32)__popcll(zkw_ballot(true));
        } else {
          any = 0;
          total = 0;
#pragma unroll
          for (u32 r = 0; r < ZKW_REGISTERS_COUNT + 1; r++) {
            const u32 c = (u32)__popcll(zkw_ballot((dm >> r) & 1u));
            total += c;
            any |= c ? 1u << r : 0u;
          }
        }
        const u32 base = delta_cur;
        const bool fits = base + total <= cap_delta;  // wave-uniform: either every lane's deltas fit or none are written
        if (ZKW_UNLIKELY(ok && !fits)) lane_fail(s, ZKW_STATUS_LIMIT);
        if (ZKW_LIKELY(fits)) {
          uint4* dl = delta_base;
          u32 pos = base;
          for (u32 left = any; left; left &= left - 1u) {  // scalar loop over the registers written in this wave-cycle
            const u32 r = (u32)__ffsll((long long)left) - 1u;
            const bool has = (dm >> r) & 1u;
            const u64 part = zkw_ballot(has);
            if (has) {
              u256 v;
              if (r == ZKW_REGISTERS_COUNT) {  // (wave-uniform)
                v = u256_zero();
                v.w[0] = cfv_heap_bound(sh, s); v.w[1] = cfv_aux_bound(sh, s); v.w[2] = s.depth;
              } else {
                v = rf_get(rf, r + 1u);
              }
              const u32 at = pos + zkw_rank_below(part);
              // two planes (low / high 16 bytes) so that each store instruction covers whole 64-byte lines
              zkw_stream_store(dl + (u64)at, u256_lo4(v));
              zkw_stream_store(dl + (u64)cap_delta + at, u256_hi4(v));
            }
            pos += (u32)__popcll(part);
          }
        }
        if (ZKW_LIKELY(ok && fits)) {
          const u32 cnt = s.counts >> 8;  // memory queries | log queries << 8 | aux events << 16 (saturating bytes)
          // dirty mask: bits 0-7 in the tail's reserved byte, bits 8-14 in the top byte of the event counts
          uint4* const tail_ptr = tails_wave + (u64)k * tail_step + s.lane;
          zkw_stream_store(tail_ptr, make_uint4((s.ptr_bitmap & 0xffffu) | ((s.flags & 0xfu) << 16) | ((dm & 0xffu) << 24),
                                                (s.pc & 0xffffu) | (s.sp << 16), s.ergs, cnt | ((dm >> 8) << 24)));
          s.kflags &= ~KF_TAIL2;
        }
        if (fits && total) {
          // `total` is the same for every lane still in the loop (ballots over exactly those lanes); the LDS copy is
          // only read after the loop (final directory entry), and a wave's LDS operations complete in order
          delta_cur = base + total;
          zkw_cursor_set<3>(delta_cur);
        }
      }
      ZKW_PROF(3)  // CycleRecord: delta ranks, delta + tail stores
#ifdef ZKW_WIDE
      if ((A.debug_flags & ZKW_DQ_HELPER) && zkw_ballot((s.kflags & KF_DQ_CHAINED) != 0)) {  // (wave-uniform; rare: a far call with a decommit)
        // hand this cycle's decommits to the helper wave: every lane still in the loop marks its entry of the slot valid
        // (its cycle completed with a decommit) or empty, then the slot is posted — a wave's LDS operations complete in
        // order, so the helper that sees the new count sees the entries
        const u32 area = dq_helper_area(sh);
        const u32 posted = *ZKW_LDS_AT(area);
        const u32 row = area + 16u + (posted & 1u) * 768u + zkw_lane_id() * 4u;
        const bool valid = (s.kflags & KF_DQ_CHAINED) && lane_ok(s);
        const u32 w0 = *ZKW_LDS_AT(row);
        *ZKW_LDS_AT(row) = valid ? (w0 | 0x80000000u) : 0u;
        if (zkw_rank_below(zkw_ballot(1)) == 0) *ZKW_LDS_AT(area) = posted + 1u;
      }
#endif
      k++;
      dir_ptr += 4;
      // leave: failed / out of cycles / execution_has_ended() (mod.rs:96-98: callers stop cycling at depth 0)
      if (ZKW_UNLIKELY(!lane_ok(s) || k >= run_cycles || s.depth == 0)) {
#ifdef ZKW_SHORT_STATS
        if (blockIdx.x == 1 && threadIdx.x == 0) {
          printf("ZKWSHORT cycles %u short %u odd %u heavy %u bad %u\n", k, zs_short, zs_odd, zs_heavy, zs_bad);
          printf("ZKWSHORT general light %u x %llu, general UMA %u x %llu, general heavy %u x %llu, short ALU %u x %llu, short UMA %u x %llu, short mul %u x %llu clocks\n", zs_n0, zs_t0 / (zs_n0 ? zs_n0 : 1), zs_n1,
                 zs_t1 / (zs_n1 ? zs_n1 : 1), zs_n2, zs_t2 / (zs_n2 ? zs_n2 : 1), zs_n3, zs_t3 / (zs_n3 ? zs_n3 : 1), zs_n4, zs_t4 / (zs_n4 ? zs_n4 : 1), zs_n5, zs_t5 / (zs_n5 ? zs_n5 : 1));
        }
#endif
        if (!lane_ok(s)) dq_undo(P, sh, s);  // the failed cycle leaves no records: a decommit it chained inline goes too
        lane_writeback(P, sh, rf, s, lane_ok(s) ? k : k - 1u);  // a lane that failed did not complete its last cycle
        break;
      }
    }
  }
#ifdef ZKW_WIDE
  if (A.helpers && tid == 0) *ZKW_LDS_AT(dq_helper_area(sh) + 8u) = 1u;  // every lane has left the loop: nothing more will be posted (before the barriers of the profiling builds below: the helper waves leave on it)
#endif
#ifdef ZKW_PROFILE
  __syncthreads();
  if (blockIdx.x == 1 && threadIdx.x == 0) {
    printf("ZKWPROF cycles %u: fetch %llu select %llu eoc %llu record %llu\n", k, zp_acc[0][0], zp_acc[0][1], zp_acc[0][2], zp_acc[0][3]);
    for (int o = 0; o < 16; o++)
      if (zp_acc[0][24 + o]) printf("ZKWPROF opcode %d: %llu iterations, %llu clocks each\n", o, zp_acc[0][24 + o], zp_acc[0][8 + o] / zp_acc[0][24 + o]);
    for (int o = 40; o < 80; o++)
      if (zp_acc[0][o] && o != 63) printf("ZKWPROF sub %d: %llu clocks in total\n", o, zp_acc[0][o]);
  }
#endif
#ifdef ZKW_WAITPROF
  __syncthreads();
  if (blockIdx.x == 1 && threadIdx.x == 0) {
    for (int o = 0; o < 16; o++)
      if (zw_acc[0][16 + o]) printf("ZKWWAIT site %d: %llu waits, %llu clocks in total\n", o, zw_acc[0][16 + o], zw_acc[0][o]);
  }
#endif
  // wave-cycles executed = the maximum over the lanes (lanes leave the loop at different iterations)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const u32 o = (u32)__shfl_xor((int)k, off);
    k = o > k ? o : k;
  }
  dir_ptr = P.dir + ((u64)wave * (P.max_cycles + 1) + cycle_base + k) * 4;
  // final directory entry
  {
    const uint4 fin = make_uint4(zkw_cursor_get<0>(), zkw_cursor_get<1>(), zkw_cursor_get<2>(), zkw_cursor_get<3>());
    if (tid == 0) {
      *(uint4*)dir_ptr = fin;
      *(uint4*)(P.cursors + wave * 4) = fin;
    }
  }
  if (tid == 0) P.wave_cycles[wave] = cycle_base + k;
}

// working state := pristine images (register file, scalars, callstack, frame meta, storage table, heap
// image) and stream cursors := 0 — one launch instead of seven copies per reset
__global__ void zkw_reset_kernel(zkw_fused_table T) {
  const zkw_reset_params ZKW_CONST_AS& R = *(const zkw_reset_params ZKW_CONST_AS*)T.p[blockIdx.y];
  const bool first = T.reserved[1] != 0;  // first reset after an upload: everything is copied
  const u32 skip = T.reserved[2];         // profiling ablation (ZKW_RESET_SKIP): parts left out
  // Every part below is a short chain of dependent memory round trips per thread (mask -> words -> stores), each several
  // microseconds on cold pages, so the parts run side by side in disjoint ranges of the grid instead of one after the
  // other in every thread: 1/8 of the workgroups copy the flat images and clear the small state, 1/8 restore storage
  // slots, 3/4 restore heap words (one dirty mask per thread at 4096 instances).  (A grid of fewer than 8 workgroups —
  // the emulation build launches one thread — does everything in every thread.)
  const u32 nb = gridDim.x, e8 = nb / 8;
  u32 part = 3, pb = blockIdx.x, pn = nb;
  if (nb >= 8) {
    if (blockIdx.x < e8) { part = 0; pn = e8; }
    else if (blockIdx.x < 2 * e8) { part = 1; pb = blockIdx.x - e8; pn = e8; }
    else { part = 2; pb = blockIdx.x - 2 * e8; pn = nb - 2 * e8; }
  }
  const u32 stride = pn * blockDim.x;
  const u32 t0 = pb * blockDim.x + threadIdx.x;
  const bool copies = part == 3 || part == 0, storage = part == 3 || part == 1, heap = part == 3 || part == 2;
  // four independent 16-byte loads in flight per thread before the stores (a latency-bound copy otherwise)
  if (copies) {
#pragma unroll 1
    for (int b = (skip & 1u) ? 5 : 0; b < (first ? 5 : 4); b++) {
      const uint4* src = R.src[b];
      uint4* dst = R.dst[b];
      const u32 n = R.n16[b];
      u32 i = t0;
      for (; i + 3 * stride < n; i += 4 * stride) {
        const uint4 a = src[i], c = src[i + stride], d = src[i + 2 * stride], e = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = c; dst[i + 2 * stride] = d; dst[i + 3 * stride] = e;
      }
      for (; i < n; i += stride) dst[i] = src[i];
      if (b == 1) {  // callstack next: only the entries a run can read before writing them (0 .. initial depth) are restored
        const u32 rows = R.n16[2], r16 = R.cs_row16, p16 = R.cs_pitch16;
        const uint4* cs = R.src[2];
        uint4* cd = R.dst[2];
        u32 j = t0;
        for (; j + 3 * stride < rows; j += 4 * stride) {
          const u32 j1 = j + stride, j2 = j + 2 * stride, j3 = j + 3 * stride;
          const u32 a0 = (j / r16) * p16 + j % r16, a1 = (j1 / r16) * p16 + j1 % r16, a2 = (j2 / r16) * p16 + j2 % r16, a3 = (j3 / r16) * p16 + j3 % r16;
          const uint4 v0 = cs[a0], v1 = cs[a1], v2 = cs[a2], v3 = cs[a3];
          cd[a0] = v0; cd[a1] = v1; cd[a2] = v2; cd[a3] = v3;
        }
        for (; j < rows; j += stride) {
          const u32 at = (j / r16) * p16 + j % r16;
          cd[at] = cs[at];
        }
        b = 2;  // skip the flat copy of buffer 2
      }
    }
  }
  if (first) {
    if (storage)
      for (u32 i = t0; i < R.n_instances * ((R.storage_slots + 31u) >> 5); i += stride) R.storage_dirty[i] = 0;
  } else if (storage && !(skip & 2u)) {
    // storage table: only the slots the run wrote (claimed, marked warm, written) — one thread per 32-slot mask.  The
    // masks are read-only here (the first launch after a reset clears them: zkw_cycle_kernel), and the six 16-byte units
    // of a slot are loaded before any is stored: one dependent memory round trip per dirty slot, not twelve.
    const u32 e16 = (u32)(sizeof(zkw_dev_storage_entry) / 16);
    const u32 sw = (R.storage_slots + 31u) >> 5;  // mask words per instance: one bit per slot
    for (u32 i = t0; i < R.n_instances * sw; i += stride) {
      u32 m = R.storage_dirty[i];
      const u64 first_slot = (u64)(i / sw) * R.storage_slots + (i % sw) * 32u;
      while (m) {
        const u32 bit = (u32)__ffsll((long long)m) - 1u;
        m &= m - 1u;
        const u64 at = (first_slot + bit) * e16;
        uint4 v[sizeof(zkw_dev_storage_entry) / 16];
#pragma unroll
        for (u32 k = 0; k < e16; k++) v[k] = R.src[4][at + k];
#pragma unroll
        for (u32 k = 0; k < e16; k++) R.dst[4][at + k] = v[k];
      }
    }
  }
  // heap image: [n_waves][heap_row16] (dense) -> rows of the working arena
  const u32 row = R.heap_row16;
  if (row && first) {  // first reset after an upload: the whole image, one flat index space
    if (heap) {
      const u32 total = R.n_waves * row;
      u32 i = t0;
      for (; i + 3 * stride < total; i += 4 * stride) {
        const u32 i1 = i + stride, i2 = i + 2 * stride, i3 = i + 3 * stride;
        const uint4 a = R.heap_src[i], c = R.heap_src[i1], d = R.heap_src[i2], e = R.heap_src[i3];
        R.heap_dst[(u64)(i / row) * R.heap_pitch16 + i % row] = a;
        R.heap_dst[(u64)(i1 / row) * R.heap_pitch16 + i1 % row] = c;
        R.heap_dst[(u64)(i2 / row) * R.heap_pitch16 + i2 % row] = d;
        R.heap_dst[(u64)(i3 / row) * R.heap_pitch16 + i3 % row] = e;
      }
      for (; i < total; i += stride) R.heap_dst[(u64)(i / row) * R.heap_pitch16 + i % row] = R.heap_src[i];
      const u32 nd = R.n_waves * ((R.image_words + 31u) >> 5) * R.L;
      for (u32 j = t0; j < nd; j += stride) R.heap_dirty[j] = 0;
    }
  } else if (row && heap && !(skip & 4u)) {
    // later resets: only the words the run overwrote (the cycle kernel sets one bit per overwritten image word; the
    // first launch after a reset clears the masks, here they are read-only).  One thread owns one 32-word mask of one
    // lane and restores four words per round — eight loads in flight, then eight stores; a mask with fewer words left
    // repeats its last word (the same bytes are written twice).  One word per round is a chain of dependent memory
    // round trips: at ~50 overwritten words per cfg-2 instance that chain was most of the 105 us this kernel took.
    const u32 groups = (R.image_words + 31u) >> 5;
    const u32 nd = R.n_waves * groups * R.L;
    for (u32 j = t0; j < nd; j += stride) {
      u32 m = R.heap_dirty[j];
      if (!m) continue;
      const u32 lane = j % R.L, g = (j / R.L) % groups, w = j / (R.L * groups);
      const uint4* src = R.heap_src + (u64)w * row + lane;
      uint4* dst = R.heap_dst + (u64)w * R.heap_pitch16 + lane;
      const u32 two_l = 2u * R.L, base = g * 32u;
      while (m) {
        u32 o[4], last = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (m) {
            last = (base + (u32)__ffsll((long long)m) - 1u) * two_l;  // [word][2][lane] inside the wave's row
            m &= m - 1u;
          }
          o[q] = last;
        }
        const uint4 a0 = src[o[0]], c0 = src[o[0] + R.L], a1 = src[o[1]], c1 = src[o[1] + R.L];
        const uint4 a2 = src[o[2]], c2 = src[o[2] + R.L], a3 = src[o[3]], c3 = src[o[3] + R.L];
        dst[o[0]] = a0; dst[o[0] + R.L] = c0; dst[o[1]] = a1; dst[o[1] + R.L] = c1;
        dst[o[2]] = a2; dst[o[2] + R.L] = c2; dst[o[3]] = a3; dst[o[3] + R.L] = c3;
      }
    }
  }
  if (copies) {
    if (R.commit_out && !(skip & 8u))
      for (u32 i = t0; i < R.n_instances; i += stride) {
        u64* tl = R.commit_out + ((u64)i * ZKW_QUEUE_COUNT + ZKW_QUEUE_DECOMMIT) * 4;
        tl[0] = tl[1] = tl[2] = tl[3] = 0;
        R.dq_count[i] = 0;
      }
    for (u32 i = t0; i < R.n_waves * 4; i += stride) R.cursors[i] = 0;
    for (u32 i = t0; i < R.n_waves; i += stride) R.wave_cycles[i] = 0;
    for (u32 i = t0; i < R.history16; i += stride) R.history[i] = make_uint4(0, 0, 0, 0);  // SimpleDecommitter starts with an empty history
  }
}

extern "C" hipError_t zkw_launch_reset_kernel(const zkw_fused_table* T, hipStream_t stream) {
  const u32 threads = T->wave_threads > 1 ? 256 : 1;
  // ~20 workgroups per CU in total, however many batches share the launch (3/4 of them restore heap words: one dirty
  // mask per thread at 4096 instances and 20 batches)
  u32 blocks = T->wave_threads > 1 ? (5120 + T->n - 1) / T->n : 1;
  if (blocks < 64 && T->wave_threads > 1) blocks = 64;
  hipLaunchKernelGGL(zkw_reset_kernel, dim3(blocks, T->n), dim3(threads), 0, stream, *T);
  return hipGetLastError();
}

// dynamic LDS per workgroup: ISA table + per wave (cursors + per-lane cold state and previous_code_word)
extern "C" uint32_t zkw_cycle_kernel_lds_bytes(uint32_t L, uint32_t waves_per_group) {
  (void)L;  // rows have a fixed lane stride
  return ZKW_ISA_TABLE_SIZE * 8 + ZKW_LDS_SINK_UNITS * 16 + waves_per_group * (32 + ZKW_LDS_STRIDE * (ZKW_COLD_FIELDS * 4 + 64 + 16 + 32));
}

// host-callable launcher (keeps <<<>>> out of the runtime)
extern "C" hipError_t zkw_launch_cycle_kernel(const zkw_launch_args* A, hipStream_t stream) {
  const u32 g = A->waves_per_group;
  const uint32_t lds = zkw_cycle_kernel_lds_bytes(A->max_L, g) + (A->helpers ? g * (ZKW_DQ_HELPER_BYTES + ((A->debug_flags & ZKW_KECCAK_HELPER) ? ZKW_KH_BYTES : 0u)) : 0u);
  zkw_launch_args args = *A;
  args.lds_sink = ZKW_ISA_TABLE_SIZE * 8;  // right behind the ISA table
  A = &args;
  if (lds > 64u * 1024u) {
    // dynamic LDS above the 64 KB default needs an explicit opt-in (not reached by the current layout: 41 KB per workgroup).  The
    // attribute is per device: remember the opted-in size per device, under a lock (contexts on several devices and
    // launches from several host threads share this function).
    static std::mutex mu;
    static uint32_t opted[64] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || lds > opted[dev]) {
      e = hipFuncSetAttribute((const void*)zkw_cycle_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) opted[dev] = lds;
    }
  }
  hipLaunchKernelGGL(zkw_cycle_kernel, dim3((A->wave_base[A->n_batches] + g - 1) / g), dim3(A->wave_threads * (g + A->helpers)), lds, stream, *A);
  return hipGetLastError();
}
