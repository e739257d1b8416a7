// TEST INFRASTRUCTURE ONLY — never part of the product, never loaded by the package.
//
// A SIMT engine for the CPU emulation build of tests/emu (ZKW_EMU_WAVE > 1): every thread of a workgroup is a fiber
// with its own stack that runs the UNMODIFIED kernel source; cross-lane operations (ballot, readlane, readfirstlane,
// shuffles, ds_bpermute, the wave's scalar cursor registers) and barriers are rendezvous points of the fibers of a
// wave / a workgroup.  What the hardware gets from its execution mask — which lanes take part in a cross-lane
// operation — comes from explicit divergence scopes in the source (ZKW_DIV_IF / ZKW_DIV_SCOPE, no-ops in the device
// build): a scope splits the wave's active set like the exec mask, runs the taken side, then the other side, and
// re-converges at its end (also for lanes that leave it early with break / continue / return).  Inside a converged
// region every active lane must arrive at the SAME cross-lane operation (same source line, same call site): anything
// else — a divergent region that holds a cross-lane operation and is not annotated — is reported and the process
// aborts, so a missing annotation cannot go unnoticed.
//
// Lanes of a wave run one after the other between two rendezvous points (not in lockstep): code that relies on
// lockstep through shared memory without a cross-lane operation in between would behave differently here; the wave's
// stream cursors — the one such place in the cycle kernel — are engine registers with a collective fetch-and-add.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <cstdio>
#include <mutex>
#include <vector>

#if ZKW_EMU_WAVE > 1

extern "C" void zkw_emu_switch(void** save_sp, void* load_sp);
// System V x86-64: callee-saved rbx, rbp, r12-r15 + the stack pointer are the whole context of a cooperative switch
asm(R"(
.text
.globl zkw_emu_switch
.type zkw_emu_switch,@function
zkw_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size zkw_emu_switch,.-zkw_emu_switch
)");

namespace {

constexpr unsigned W = ZKW_EMU_WAVE;
constexpr size_t STACK_BYTES = 1u << 20;

enum LaneState { LS_RUN, LS_COLL, LS_ELSE_WAIT, LS_PARKED, LS_DONE };

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
};

struct Frame {
  uint64_t saved_active, else_lanes, parked;
  int phase;
  const char* file;
  int line;
};

struct WaveState {
  uint64_t live = 0, active = 0, arrived = 0;
  std::vector<Frame> frames;
  int state[64];
  int kind[64];
  uint64_t operand[64];
  uint32_t arg[64];
  const char* file[64];
  int line[64];
  void* ra[64];
  uint64_t result[64];
  bool at_barrier = false, yielded = false;
  uint32_t sregs[16];
};

std::vector<Fiber> g_fibers;  // pool: fiber t of the running workgroup
std::vector<WaveState> g_waves;
void* g_sched_sp = nullptr;
unsigned g_cur = 0;  // thread of the workgroup that is running
unsigned g_nthreads = 0;
void (*g_entry)(void*) = nullptr;
void* g_entry_arg = nullptr;
std::mutex g_launch_mu;
unsigned long long g_spins = 0;

const char* kind_name(int k) {
  static const char* n[] = {"ballot", "readfirstlane", "readlane", "shfl", "shfl_xor", "bpermute", "wave_barrier", "fetch_add", "yield", "scope_begin", "syncthreads"};
  return k >= 0 && k <= 10 ? n[k] : "?";
}

[[noreturn]] void die(const char* what) {
  fprintf(stderr, "\n[zkw emu SIMT] %s (workgroup %u,%u,%u)\n", what, blockIdx.x, blockIdx.y, blockIdx.z);
  for (size_t w = 0; w < g_waves.size(); w++) {
    const WaveState& ws = g_waves[w];
    if (!ws.live) continue;
    fprintf(stderr, "  wave %zu: live %016llx active %016llx arrived %016llx%s\n", w, (unsigned long long)ws.live, (unsigned long long)ws.active, (unsigned long long)ws.arrived,
            ws.at_barrier ? " [at barrier]" : "");
    for (size_t f = 0; f < ws.frames.size(); f++)
      fprintf(stderr, "    scope %zu: %s:%d phase %d saved %016llx else %016llx parked %016llx\n", f, ws.frames[f].file, ws.frames[f].line, ws.frames[f].phase,
              (unsigned long long)ws.frames[f].saved_active, (unsigned long long)ws.frames[f].else_lanes, (unsigned long long)ws.frames[f].parked);
    // distinct sites of the blocked active lanes
    uint64_t seen = 0;
    for (unsigned l = 0; l < 64; l++) {
      if (!((ws.active >> l) & 1) || ((seen >> l) & 1)) continue;
      uint64_t same = 0;
      for (unsigned m = l; m < 64; m++)
        if (((ws.active >> m) & 1) && ws.state[m] == ws.state[l] && (ws.state[l] != LS_COLL || (ws.kind[m] == ws.kind[l] && ws.line[m] == ws.line[l] && ws.file[m] == ws.file[l]))) same |= 1ull << m;
      seen |= same;
      if (ws.state[l] == LS_COLL)
        fprintf(stderr, "    lanes %016llx: at %s %s:%d (call site %p)\n", (unsigned long long)same, kind_name(ws.kind[l]), ws.file[l], ws.line[l], ws.ra[l]);
      else
        fprintf(stderr, "    lanes %016llx: state %d\n", (unsigned long long)same, ws.state[l]);
    }
  }
  fflush(stderr);
  abort();
}

inline WaveState& cur_wave() { return g_waves[g_cur / W]; }
inline unsigned cur_lane() { return g_cur % W; }

void yield_to_scheduler() {
  Fiber& f = g_fibers[g_cur];
  zkw_emu_switch(&f.sp, g_sched_sp);
}

void fiber_main() {
  g_entry(g_entry_arg);
  WaveState& ws = cur_wave();
  const uint64_t bit = 1ull << cur_lane();
  ws.live &= ~bit; ws.active &= ~bit; ws.arrived &= ~bit;
  for (Frame& fr : ws.frames) {
    fr.saved_active &= ~bit; fr.else_lanes &= ~bit; fr.parked &= ~bit;
  }
  ws.state[cur_lane()] = LS_DONE;
  yield_to_scheduler();
  die("a finished fiber was resumed");
}

void init_fiber(Fiber& f) {
  if (!f.stack) {
    void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) die("mmap of a fiber stack failed");
    f.stack = (char*)p;
  }
  uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;               // return address of fiber_main (never used)
  *--sp = (void*)&fiber_main;    // popped by the `ret` of zkw_emu_switch
  for (int i = 0; i < 6; i++) *--sp = nullptr;  // rbp, rbx, r12..r15
  f.sp = (void*)sp;
}

void release_all(WaveState& ws, uint64_t lanes) {
  for (unsigned l = 0; l < W; l++)
    if ((lanes >> l) & 1) ws.state[l] = LS_RUN;
}

// the active lanes of a wave are all blocked at a cross-lane operation: check that it is ONE operation and complete it
void complete_collective(WaveState& ws) {
  const unsigned first = (unsigned)__builtin_ctzll(ws.active);
  for (unsigned l = first; l < W; l++)
    if (((ws.active >> l) & 1) && (ws.kind[l] != ws.kind[first] || ws.line[l] != ws.line[first] || ws.file[l] != ws.file[first]))  // (not the return address: the compiler duplicates tails)
      die("the active lanes of a wave wait at DIFFERENT cross-lane operations: a divergent region holds one and is not a ZKW_DIV_IF / ZKW_DIV_SCOPE");
  const int kind = ws.kind[first];
  const uint64_t act = ws.active;
  ws.arrived = 0;
  switch (kind) {
    case ZE_BALLOT: {
      uint64_t m = 0;
      for (unsigned l = 0; l < W; l++)
        if (((act >> l) & 1) && ws.operand[l]) m |= 1ull << l;
      for (unsigned l = 0; l < W; l++)
        if ((act >> l) & 1) ws.result[l] = m;  // (only the active lanes: a lane that waits in a scope holds ITS result there)
      break;
    }
    case ZE_READFIRST:
      for (unsigned l = 0; l < W; l++)
        if ((act >> l) & 1) ws.result[l] = ws.operand[first];
      break;
    case ZE_READLANE:
    case ZE_SHFL:
      for (unsigned l = 0; l < W; l++)
        if ((act >> l) & 1) {
          const unsigned src = ws.arg[l] & (W - 1);
          ws.result[l] = ((act >> src) & 1) ? ws.operand[src] : 0;  // (an inactive lane's register: not modelled, reads 0)
        }
      break;
    case ZE_SHFL_XOR:
      for (unsigned l = 0; l < W; l++)
        if ((act >> l) & 1) {
          const unsigned src = (l ^ ws.arg[l]) & (W - 1);
          ws.result[l] = ((act >> src) & 1) ? ws.operand[src] : ws.operand[l];
        }
      break;
    case ZE_BPERMUTE:
      for (unsigned l = 0; l < W; l++)
        if ((act >> l) & 1) {
          const unsigned src = (ws.arg[l] >> 2) & (W - 1);
          ws.result[l] = ((act >> src) & 1) ? ws.operand[src] : 0;
        }
      break;
    case ZE_WAVE_BARRIER:
      break;
    case ZE_YIELD:
      ws.yielded = true;
      break;
    case ZE_FETCH_ADD: {
      const uint32_t r = ws.arg[first] & 15u;
      const uint32_t old = ws.sregs[r];
      ws.sregs[r] = old + (uint32_t)ws.operand[first];
      for (unsigned l = 0; l < W; l++)
        if ((act >> l) & 1) ws.result[l] = old;
      break;
    }
    case ZE_SCOPE_BEGIN: {
      Frame fr;
      fr.saved_active = act;
      fr.parked = 0;
      fr.file = ws.file[first];
      fr.line = ws.line[first];
      uint64_t taken = 0;
      for (unsigned l = 0; l < W; l++)
        if (((act >> l) & 1) && ws.operand[l]) taken |= 1ull << l;
      fr.else_lanes = act & ~taken;
      for (unsigned l = 0; l < W; l++)
        if ((act >> l) & 1) ws.result[l] = (taken >> l) & 1;
      if (taken) {
        fr.phase = 1;
        ws.active = taken;
        for (unsigned l = 0; l < W; l++)
          if ((fr.else_lanes >> l) & 1) ws.state[l] = LS_ELSE_WAIT;
      } else {
        fr.phase = 2;
        ws.active = fr.else_lanes;
        fr.else_lanes = 0;
      }
      ws.frames.push_back(fr);
      release_all(ws, ws.active);
      return;
    }
    case ZE_SYNCTHREADS:
      ws.at_barrier = true;  // stays blocked: released by the scheduler when every live wave of the workgroup is here
      ws.arrived = act;
      return;
    default: die("unknown cross-lane operation");
  }
  release_all(ws, act);
}

// settles what can be settled without running a lane; true if a lane became runnable
bool resolve(WaveState& ws) {
  bool progress = false;
  for (;;) {
    if (!ws.live) return progress;
    if (ws.active == 0) {
      if (ws.frames.empty()) die("a wave has live lanes but none active and no open scope");
      Frame& fr = ws.frames.back();
      if (fr.phase == 1 && (fr.else_lanes & ws.live)) {
        ws.active = fr.else_lanes & ws.live;
        fr.else_lanes = 0;
        fr.phase = 2;
        release_all(ws, ws.active);
        return true;
      }
      const uint64_t back = fr.saved_active & ws.live;
      ws.frames.pop_back();
      ws.active = back;
      release_all(ws, back);
      if (back) return true;
      progress = true;
      continue;  // every lane of that scope has finished: the scope around it
    }
    if ((ws.arrived & ws.active) == ws.active && !ws.at_barrier) {
      complete_collective(ws);
      if (ws.at_barrier) return progress;
      progress = true;
      if (ws.active) return true;
      continue;
    }
    return progress;
  }
}

void run_block(unsigned nthreads) {
  g_nthreads = nthreads;
  if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
  const unsigned nw = (nthreads + W - 1) / W;
  g_waves.assign(nw, WaveState());
  for (unsigned t = 0; t < nthreads; t++) {
    init_fiber(g_fibers[t]);
    WaveState& ws = g_waves[t / W];
    ws.live |= 1ull << (t % W);
    ws.state[t % W] = LS_RUN;
  }
  for (WaveState& ws : g_waves) {
    ws.active = ws.live;
    for (unsigned i = 0; i < 16; i++) ws.sregs[i] = 0;
  }
  g_spins = 0;
  for (;;) {
    bool progress = false, any_live = false;
    for (unsigned w = 0; w < nw; w++) {
      WaveState& ws = g_waves[w];
      if (!ws.live) continue;
      any_live = true;
      ws.yielded = false;
      for (;;) {  // this wave as far as it gets on its own
        bool ran = false;
        for (unsigned l = 0; l < W; l++) {
          if (!((ws.active >> l) & 1) || ws.state[l] != LS_RUN) continue;
          g_cur = w * W + l;
          threadIdx = dim3(g_cur);
          zkw_emu_switch(&g_sched_sp, g_fibers[g_cur].sp);
          ran = true;
        }
        const bool settled = resolve(ws);
        if (ran || settled) progress = true;
        if (!(ran || settled) || ws.yielded || !ws.live || ws.at_barrier) break;
      }
    }
    if (!any_live) break;
    // workgroup barrier: every wave that still has live lanes waits at it
    bool all_at = true, some_at = false;
    for (WaveState& ws : g_waves)
      if (ws.live) {
        all_at = all_at && ws.at_barrier;
        some_at = some_at || ws.at_barrier;
      }
    if (some_at && all_at) {
      for (WaveState& ws : g_waves)
        if (ws.live) {
          ws.at_barrier = false;
          ws.arrived = 0;
          release_all(ws, ws.active);
        }
      progress = true;
    }
    if (!progress) die("deadlock: no lane of the workgroup can run");
    if (++g_spins > 200000000ull) die("livelock: the workgroup keeps yielding");
  }
}

}  // namespace

extern "C" void zkw_emu_launch(void (*entry)(void*), void* arg, dim3 grid, dim3 block) {
  std::lock_guard<std::mutex> lock(g_launch_mu);
  g_entry = entry;
  g_entry_arg = arg;
  gridDim = grid;
  blockDim = block;
  const unsigned nthreads = block.x * block.y * block.z;
  if (block.y != 1 || block.z != 1) die("only one-dimensional workgroups are emulated");
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned b = 0; b < grid.x; b++) {
        blockIdx = dim3(b, by, bz);
        run_block(nthreads);
      }
}

extern "C" uint64_t zkw_emu_collective(int kind, uint64_t operand, uint32_t arg, const char* file, int line) {
  WaveState& ws = cur_wave();
  const unsigned l = cur_lane();
  ws.kind[l] = kind; ws.operand[l] = operand; ws.arg[l] = arg; ws.file[l] = file; ws.line[l] = line;
  ws.ra[l] = __builtin_return_address(0);
  ws.state[l] = LS_COLL;
  ws.arrived |= 1ull << l;
  yield_to_scheduler();
  return cur_wave().result[cur_lane()];
}

extern "C" void zkw_emu_scope_end(void) {
  WaveState& ws = cur_wave();
  const unsigned l = cur_lane();
  if (ws.frames.empty()) die("end of a divergence scope without an open scope");
  ws.frames.back().parked |= 1ull << l;
  ws.active &= ~(1ull << l);
  ws.state[l] = LS_PARKED;
  yield_to_scheduler();
}

extern "C" uint32_t* zkw_emu_wave_sregs(void) { return cur_wave().sregs; }

#endif  // ZKW_EMU_WAVE > 1
