"""Synthetic workloads of SURVEY.md §8d / BASELINE.json `configs`.

Every instance runs in kernel mode (`this_address = 0x8001`) so that fat-pointer metadata
erasure (reference cycle.rs:381) and the privilege check (cycle.rs:174) never fire, starts with
2^32-1 ergs, and shares one opcode tape; register files, heaps and storage are per instance.
Randomness: xoshiro256** streams seeded from `0x5eed0000 + cfg` through splitmix64.

  cfg 0   1 x 1024      alternating NOP / ADD r1,r2->r3                      (CPU plumbing)
  cfg 1   256 x 256     ADD/SUB/MUL/DIV reg-reg, 1/64 DIV by r0, 50% set_flags, 1/8 predicated
  cfg 2   4096 x 256    ALU + unaligned heap ld/st + stack operands + jumps + 2 far_call->ret
                        pairs (2 fresh decommits of 512-word bytecodes) per instance
"""
import hashlib

import numpy as np

from . import capi as K

M64 = (1 << 64) - 1


# ----------------------------------------------------------------------------------------
# xoshiro256** (vectorised over independent streams)
# ----------------------------------------------------------------------------------------
def _splitmix(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return x, z


class Xoshiro:
    def __init__(self, seed, n_streams=1):
        x = (np.uint64(seed) + np.arange(n_streams, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)).astype(np.uint64)
        s = []
        for _ in range(4):
            x, z = _splitmix(x)
            s.append(z)
        self.s = s

    def next(self):
        s0, s1, s2, s3 = self.s
        with np.errstate(over="ignore"):
            r = s1 * np.uint64(5)
            r = ((r << np.uint64(7)) | (r >> np.uint64(57))) * np.uint64(9)
            t = s1 << np.uint64(17)
            s2 = s2 ^ s0
            s3 = s3 ^ s1
            s1 = s1 ^ s2
            s0 = s0 ^ s3
            s2 = s2 ^ t
            s3 = (s3 << np.uint64(45)) | (s3 >> np.uint64(19))
        self.s = [s0, s1, s2, s3]
        return r

    def words(self, n_words):
        """[n_streams, n_words, 4] u64"""
        out = np.empty((len(self.s[0]), n_words, 4), dtype="<u8")
        for w in range(n_words):
            for l in range(4):
                out[:, w, l] = self.next()
        return out


class ScalarRng:
    def __init__(self, seed):
        self.x = Xoshiro(seed, 1)

    def u64(self):
        return int(self.x.next()[0])

    def below(self, n):
        return self.u64() % n

    def chance(self, num, den):
        return self.below(den) < num


# ----------------------------------------------------------------------------------------
# workload container
# ----------------------------------------------------------------------------------------
class Workload:
    def __init__(self, name, n_instances, n_cycles):
        self.name = name
        self.n_instances = n_instances
        self.n_cycles = n_cycles
        self.blobs = []       # list of [n_words, 4] u64
        self.preimages = []   # (hash[4] u64, blob index)
        self.code_pages = []  # (first, count, page, blob index)
        self.states = None
        self.inner = None
        self.heaps = None     # [n, words, 4] u64
        self.storage = None   # list of STORAGE_SLOT arrays
        self.default_aa_code_hash = np.zeros(4, dtype="<u8")
        self.zkporter_is_available = 0
        self.limits = dict(max_cycles=n_cycles, max_far_frames=4, max_callstack_depth=8, stack_words=128, heap_words=512, aux_heap_words=64,
                           storage_slots=16, storage_journal=16, max_mem_queries=0, max_log_queries=0, max_aux_events=0, lanes_per_wave=0)


BOOTLOADER_BASE_PAGE = 8
BOOTLOADER_CODE_PAGE = 8
STARTING_TIMESTAMP = 1024
KERNEL_ADDRESS = 0x8001


def initial_states(n, registers, heap_bound=4096, ergs=0xFFFFFFFF, first_dynamic_page=16):
    """VmLocalState right after `push_bootloader_context` (helpers.rs:289-316): callstack =
    [empty_context (execution_stack.rs:35-55), bootloader frame]."""
    st = np.zeros(n, dtype=K.VM_LOCAL_STATE)
    st["registers"] = registers
    st["timestamp"] = STARTING_TIMESTAMP
    st["memory_page_counter"] = first_dynamic_page
    st["callstack_depth"] = 1
    cur = st["current"]
    cur["this_address"] = K.address_bytes(KERNEL_ADDRESS)
    cur["code_address"] = K.address_bytes(KERNEL_ADDRESS)
    cur["base_memory_page"] = BOOTLOADER_BASE_PAGE
    cur["code_page"] = BOOTLOADER_CODE_PAGE
    cur["ergs_remaining"] = ergs
    cur["heap_bound"] = heap_bound
    cur["aux_heap_bound"] = heap_bound
    inner = np.zeros((n, 1), dtype=K.CALLSTACK_ENTRY)
    inner["ergs_remaining"] = 0xFFFFFFFF - ergs  # VM_INITIAL_FRAME_ERGS - passed
    return st, inner


# ----------------------------------------------------------------------------------------
# cfg 0 / cfg 1
# ----------------------------------------------------------------------------------------
def config0(isa, n_cycles=1024, seed=0x5EED0000):
    wl = Workload("cfg0_nop_add", 1, n_cycles)
    ops = []
    n_add = 0
    for k in range(n_cycles):
        if k % 2 == 0:
            ops.append(isa.enc(K.OP_NOP))
        else:
            ops.append(isa.enc(K.OP_ADD, flags=n_add % 2, src0=1, src1=2, dst0=3))
            n_add += 1
    wl.blobs.append(K.pack_code(ops))
    wl.code_pages.append((0, 1, BOOTLOADER_CODE_PAGE, 0))
    regs = Xoshiro(seed, 1).words(15)
    regs[:, 2:] = 0
    wl.states, wl.inner = initial_states(1, regs)
    return wl


def arith_tape(isa, n_cycles, rng):
    ops = []
    for k in range(n_cycles):
        which = rng.below(4)
        set_flags = rng.below(2)
        cond = (1 + rng.below(3)) if rng.chance(1, 8) else K.COND_ALWAYS
        src0 = 1 + rng.below(15)
        src1 = 1 + rng.below(15)
        dst0 = 3 + (k % 13)
        dst1 = 3 + rng.below(13)
        if which == 0:
            ops.append(isa.enc(K.OP_ADD, flags=set_flags, cond=cond, src0=src0, src1=src1, dst0=dst0))
        elif which == 1:
            ops.append(isa.enc(K.OP_SUB, flags=set_flags | (rng.below(2) << 1), cond=cond, src0=src0, src1=src1, dst0=dst0))
        elif which == 2:
            ops.append(isa.enc(K.OP_MUL, flags=set_flags, cond=cond, src0=src0, src1=src1, dst0=dst0, dst1=dst1))
        else:
            if rng.chance(1, 64):
                src1 = 0  # divisor forced to r0 == 0
            ops.append(isa.enc(K.OP_DIV, flags=set_flags, cond=cond, src0=src0, src1=src1, dst0=dst0, dst1=dst1))
    return ops


def config1(isa, n_instances=256, n_cycles=256, seed=0x5EED0001):
    wl = Workload("cfg1_arith", n_instances, n_cycles)
    ops = arith_tape(isa, n_cycles, ScalarRng(seed))
    wl.blobs.append(K.pack_code(ops))
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    regs = Xoshiro(seed ^ 0xABCDEF, n_instances).words(15)
    # a quarter of the registers are short (64..192 bit) so that DIV sees non-trivial quotients
    regs[:, 5::4, 3] = 0
    regs[:, 6::4, 2:] = 0
    wl.states, wl.inner = initial_states(n_instances, regs)
    return wl


# ----------------------------------------------------------------------------------------
# cfg 2
# ----------------------------------------------------------------------------------------
CONST_BASE = 2000  # word index of the constant pool inside every code page
HEAP_BYTES = 8192
CALLEE_CODE_WORDS = 512
ADDR_A, ADDR_B = 0x10001, 0x10002  # user-space callee addresses (>= 2^16)
CALLEE_CYCLES = 16
RELOAD_CYCLES = 14


def versioned_code_hash(words):
    """ContractCodeSha256 versioned hash (far_call.rs:169-252 consumes it): byte0 = 1,
    byte1 = 0 (at rest), bytes 2-3 = length in words (BE), rest = sha256 tail."""
    raw = words.astype(">u8")[:, ::-1].tobytes()  # 32-byte big-endian words
    digest = hashlib.sha256(raw).digest()
    be = bytes([1, 0, (len(words) >> 8) & 0xFF, len(words) & 0xFF]) + digest[4:]
    return K.u256_from_int(int.from_bytes(be, "big"))


# FarCallForwardPageType -> its ABI byte is a constant of the table the workload is built for (zkw_isa_consts.forwarding_codes):
# the table is an explicit argument (a module-level binding was overwritten by whichever constructor ran last — the metamorphic
# tests build workloads for two tables in one process)
def far_call_abi(isa, start, length, ergs_passed, forwarding_mode=0):
    """forwarding_mode: 0 UseHeap, 1 ForwardFatPointer, 2 UseAuxHeap (logical; encoded through `isa`)"""
    v = (start << 64) | (length << 96) | (ergs_passed << 192) | (isa.fwd_code(forwarding_mode) << 224)
    return K.u256_from_int(v)


def ret_abi(isa, start, length, forwarding_mode=0):
    return K.u256_from_int((start << 64) | (length << 96) | (isa.fwd_code(forwarding_mode) << 224))


class TapeBuilder:
    """Emits a straight-line program and tracks the pc so that forward jumps can be placed."""

    def __init__(self, isa, rng):
        self.isa = isa
        self.rng = rng
        self.ops = []
        self.executed = 0
        self.sp = 0
        self.k = 0

    def emit(self, op):
        self.ops.append(op)
        self.executed += 1

    def filler(self):
        r = self.rng
        return self.isa.enc(K.OP_ADD, src0=1 + r.below(12), src1=1 + r.below(12), dst0=3 + r.below(9))

    def src(self):
        return 1 + self.rng.below(12)

    def dst(self):
        self.k += 1
        return 3 + (self.k % 9)  # r3..r11; r12 is the heap cursor

    def alu(self):
        r, isa = self.rng, self.isa
        which = r.below(8)
        sf = r.below(2)
        a, b, d = self.src(), self.src(), self.dst()
        if which == 0:
            self.emit(isa.enc(K.OP_ADD, flags=sf, src0=a, src1=b, dst0=d))
        elif which == 1:
            self.emit(isa.enc(K.OP_SUB, flags=sf | (r.below(2) << 1), src0=a, src1=b, dst0=d))
        elif which == 2:
            self.emit(isa.enc(K.OP_MUL, flags=sf, src0=a, src1=b, dst0=d, dst1=3 + r.below(9)))
        elif which == 3:
            self.emit(isa.enc(K.OP_DIV, flags=sf | (r.below(2) << 1), src0=a, src1=b, dst0=d, dst1=3 + r.below(9)))
        elif which == 4:
            self.emit(isa.enc(K.OP_SHIFT, variant=r.below(4), flags=sf | (r.below(2) << 1), src0=a, src1=b, dst0=d))
        elif which == 5:
            self.emit(isa.enc(K.OP_BINOP, variant=r.below(3), flags=sf, src0=a, src1=b, dst0=d))
        elif which == 6:  # constant from the code page (UseCodePage operand)
            self.emit(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, flags=sf, src0=0, imm0=CONST_BASE + 4 + r.below(4), src1=b, dst0=d))
        else:  # 16-bit immediate operand
            self.emit(isa.enc(K.OP_SUB, src0_mode=K.MODE_IMM, flags=sf, imm0=r.below(65536), src1=b, dst0=d))

    def heap_offset(self):
        r = self.rng
        if r.below(2):
            return 32 * r.below(HEAP_BYTES // 32)
        return r.below(HEAP_BYTES)

    def heap_ld(self):
        r, isa = self.rng, self.isa
        if r.chance(1, 4):  # register-addressed with post-increment: set the cursor, then load through it
            self.emit(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_READ, src0_mode=K.MODE_REG, flags=1, src0=12, dst0=self.dst(), dst1=12))
        else:
            self.emit(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_READ, src0_mode=K.MODE_IMM, flags=0, imm0=self.heap_offset(), dst0=self.dst()))

    def heap_st(self):
        r, isa = self.rng, self.isa
        if r.chance(1, 4):
            self.emit(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_REG, flags=1, src0=12, src1=self.src(), dst0=12))
        else:
            self.emit(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, flags=0, imm0=self.heap_offset(), src1=self.src()))

    def set_cursor(self):
        # r12 := imm16 (heap cursor for the register-addressed UMA forms)
        self.emit(self.isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=self.heap_offset() & 0x1FFF, src1=0, dst0=12))

    def stack_op(self):
        r, isa = self.rng, self.isa
        choice = r.below(4)
        if self.sp == 0 or choice == 0:  # push
            self.emit(isa.enc(K.OP_ADD, dst0_mode=K.MODE_STACK_PP, src0=self.src(), src1=self.src(), dst0=0, imm1=1))
            self.sp += 1
        elif choice == 1:  # pop
            self.emit(isa.enc(K.OP_ADD, src0_mode=K.MODE_STACK_PP, src0=0, imm0=1, src1=self.src(), dst0=self.dst()))
            self.sp -= 1
        elif choice == 2:  # sp-relative read
            self.emit(isa.enc(K.OP_BINOP, variant=K.BINOP_XOR, src0_mode=K.MODE_STACK_OFF, src0=0, imm0=1 + r.below(self.sp), src1=self.src(), dst0=self.dst()))
        else:  # absolute write
            self.emit(isa.enc(K.OP_SUB, dst0_mode=K.MODE_STACK_ABS, src0=self.src(), src1=self.src(), dst0=0, imm1=r.below(self.sp)))

    def jump(self):
        skip = self.rng.below(4)
        target = len(self.ops) + 1 + skip
        self.emit(self.isa.enc(K.OP_JUMP, src0_mode=K.MODE_IMM, imm0=target))
        for _ in range(skip):
            self.ops.append(self.filler())  # never executed

    def random_segment(self, n):
        # 40% ALU, 20% heap ld, 20% heap st, 10% stack operands, 10% jumps
        self.set_cursor()
        for _ in range(n - 1):
            x = self.rng.below(10)
            if x < 4:
                if self.rng.chance(1, 6):
                    self.set_cursor()
                else:
                    self.alu()
            elif x < 6:
                self.heap_ld()
            elif x < 8:
                self.heap_st()
            elif x < 9:
                self.stack_op()
            else:
                self.jump()

    def reload(self, abi_const, dest_const):
        """After a far-call return every register but r1 is zero (ret.rs:213-233): refill r2..r11 from
        the (per-instance) heap and r13/r14 from code constants. RELOAD_CYCLES ops."""
        isa = self.isa
        for d in range(2, 12):
            self.emit(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_READ, src0_mode=K.MODE_IMM, imm0=32 * (d * 7 % 200), dst0=d))
        self.emit(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=CONST_BASE + abi_const, src1=0, dst0=13))
        self.emit(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=CONST_BASE + dest_const, src1=0, dst0=14))
        # read the first returndata word through the fat pointer in r1
        self.emit(isa.enc(K.OP_UMA, variant=K.UMA_FAT_PTR_READ, flags=0, src0=1, dst0=12))
        self.set_cursor()

    def far_call(self, variant=K.FAR_NORMAL):
        handler = len(self.ops) + 1  # exception handler = fall through
        self.emit(self.isa.enc(K.OP_FAR_CALL, variant=variant, src0=13, src1=14, imm0=handler))


def callee_program(isa, rng, ret_variant):
    """CALLEE_CYCLES executed ops: calldata reads through the fat pointer, ALU, own-heap writes,
    stack traffic, then ret with a 64-byte heap slice as returndata."""
    ops = []
    for d in (3, 4, 5):
        ops.append(isa.enc(K.OP_UMA, variant=K.UMA_FAT_PTR_READ, flags=1, src0=1, dst0=d, dst1=1))
    ops.append(isa.enc(K.OP_MUL, flags=1, src0=3, src1=4, dst0=6, dst1=7))
    ops.append(isa.enc(K.OP_ADD, flags=0, src0=5, src1=6, dst0=8))
    ops.append(isa.enc(K.OP_DIV, flags=1, src0=7, src1=5, dst0=9, dst1=10))
    ops.append(isa.enc(K.OP_SHIFT, variant=K.SHIFT_ROL, flags=0, src0=8, src1=9, dst0=11))
    ops.append(isa.enc(K.OP_BINOP, variant=K.BINOP_XOR, flags=1, src0=11, src1=3, dst0=12))
    ops.append(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=0, src1=6))
    ops.append(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=32, src1=8))
    ops.append(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=40 + rng.below(16), src1=12))
    ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=CONST_BASE + 0, src1=0, dst0=13))  # r13 := ret ABI
    ops.append(isa.enc(K.OP_ADD, dst0_mode=K.MODE_STACK_PP, src0=9, src1=10, dst0=0, imm1=1))   # push
    ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_STACK_PP, src0=0, imm0=1, src1=11, dst0=14))  # pop
    ops.append(isa.enc(K.OP_SUB, flags=1, src0=14, src1=12, dst0=15))
    ops.append(isa.enc(K.OP_RET, variant=ret_variant, flags=0, src0=13))
    assert len(ops) == CALLEE_CYCLES
    return ops


def config2(isa, n_instances=4096, n_cycles=256, seed=0x5EED0002):
    assert n_cycles >= 2 * (1 + CALLEE_CYCLES + RELOAD_CYCLES) + 12
    wl = Workload("cfg2_mixed", n_instances, n_cycles)
    rng = ScalarRng(seed)
    tb = TapeBuilder(isa, rng)
    budget = n_cycles - 2 * (1 + CALLEE_CYCLES + RELOAD_CYCLES)
    s1 = budget // 3
    s2 = budget // 3
    s3 = budget - s1 - s2
    # bootloader constants: [0]=far-call ABI A, [1]=dest A, [2]=far-call ABI B, [3]=dest B, [4..8) random words
    rnd = Xoshiro(seed ^ 0x77, 1).words(4)[0]
    consts = [far_call_abi(isa, 64, 256, 100000), K.u256_from_int(ADDR_A), far_call_abi(isa, 512, 96, 50000), K.u256_from_int(ADDR_B), rnd[0], rnd[1], rnd[2], rnd[3]]
    tb.random_segment(s1)
    tb.far_call()
    tb.executed += CALLEE_CYCLES
    tb.reload(2, 3)
    tb.random_segment(s2)
    tb.far_call()
    tb.executed += CALLEE_CYCLES
    tb.reload(0, 1)
    tb.random_segment(s3)
    assert tb.executed == n_cycles, (tb.executed, n_cycles)
    boot_words = np.zeros((CONST_BASE + len(consts), 4), dtype="<u8")
    code = K.pack_code(tb.ops)
    assert len(code) < CONST_BASE
    boot_words[: len(code)] = code
    for i, c in enumerate(consts):
        boot_words[CONST_BASE + i] = c
    wl.blobs.append(boot_words)
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    # callee bytecodes: 512 words each, constant pool in the last words
    callee_blobs = []
    for which, retv in ((0, K.RET_OK), (1, K.RET_REVERT)):
        ops = callee_program(isa, rng, retv)
        words = np.zeros((CALLEE_CODE_WORDS, 4), dtype="<u8")
        code = K.pack_code(ops)
        words[: len(code)] = code
        # filler so the preimage is not mostly zeros
        fill = Xoshiro(seed ^ (0x1000 + which), 1).words(CALLEE_CODE_WORDS - 64)[0]
        words[32 : 32 + len(fill)] = fill
        words[CALLEE_CODE_WORDS - 8 :] = 0
        callee_blobs.append((ops, words))
    # the callee's `add code[CONST_BASE + 0]` must resolve inside its own 512-word page: index 2000 > 512 reads zero => ret ABI = empty slice.
    # put the ret ABI where the callee looks for it by using a page-local constant index instead:
    for which, (ops, words) in enumerate(callee_blobs):
        local = CALLEE_CODE_WORDS - 8
        ops[11] = isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local, src1=0, dst0=13)
        code = K.pack_code(ops)
        words[: len(code)] = code
        words[local] = ret_abi(isa, 0, 64)
        wl.blobs.append(words)
        h = versioned_code_hash(words)
        wl.preimages.append((h, 1 + which))
    # storage: DEPLOYER[callee address] = versioned code hash
    slots = np.zeros(2, dtype=K.STORAGE_SLOT)
    for which, addr in enumerate((ADDR_A, ADDR_B)):
        slots[which]["key"] = K.u256_from_int(addr)
        slots[which]["value"] = wl.preimages[which][0]
        slots[which]["address"] = K.address_bytes(0x8002)
        slots[which]["shard_id"] = 0
    wl.storage = [slots] * n_instances
    regs = Xoshiro(seed ^ 0xABCDEF, n_instances).words(15)
    regs[:, 5::4, 3] = 0
    regs[:, 6::4, 2:] = 0
    regs[:, 11] = 0  # r12 = heap cursor, set by the tape
    regs[:, 12] = consts[0]  # r13 = far-call ABI A
    regs[:, 13] = consts[1]  # r14 = dest A
    regs[:, 14] = 0
    wl.states, wl.inner = initial_states(n_instances, regs)
    wl.heaps = Xoshiro(seed ^ 0x4EA9, n_instances).words(HEAP_BYTES // 32)
    wl.limits.update(max_far_frames=3, heap_words=320, stack_words=128, aux_heap_words=8, storage_slots=8, storage_journal=4)
    return wl


def make(cfg, isa, **kw):
    return {0: config0, 1: config1, 2: config2}[cfg](isa, **kw)


# ----------------------------------------------------------------------------------------
# cfg 3 — precompile-dominant trace (keccak256 / sha256 over heap data)
# ----------------------------------------------------------------------------------------
SHA256_ADDRESS = 0x02
KECCAK_ADDRESS = 0x8010
KECCAK_COST_PER_ROUND = 40
SHA256_COST_PER_ROUND = 7


def precompile_abi(in_off, in_len, out_off, out_len, page_r, page_w, extra=0):
    return K.u256_from_int(in_off | (in_len << 32) | (out_off << 64) | (out_len << 96) | (page_r << 128) | (page_w << 160) | (extra << 192))


def sha256_padded_len(n):
    return ((n + 9 + 63) // 64) * 64


def config3(isa, n_instances=4096, n_cycles=64, seed=0x5EED0003, keccak_k=(1, 8, 64, 512), sha_rounds=(1, 8, 64, 157), keccak_bytes=None,
            keccak_unalign=None):
    """keccak_bytes / keccak_unalign: optional explicit message lengths (bytes) and byte misalignments (default: 136 * k
    bytes, misalignment 0 / 31 alternating)"""
    klen = list(keccak_bytes) if keccak_bytes is not None else [136 * k for k in keccak_k]
    keccak_k = [b // 136 for b in klen]  # rounds - 1, for the ergs cost
    kun = list(keccak_unalign) if keccak_unalign is not None else [(31 if j % 2 else 0) for j in range(len(klen))]
    """The bootloader frame runs as the sha256 system contract (address 0x02) and hashes four
    regions of its own heap (precompile reads are MemoryType::Heap of the current frame), then
    far-calls the keccak system contract (0x8010) passing its whole heap as calldata; the callee
    issues four keccak256 calls that read the caller's heap page through MemoryType::FatPointer
    (2 byte-aligned + 2 with a 31-byte unaligned start) and writes the digests to its own heap."""
    wl = Workload("cfg3_precompiles", n_instances, n_cycles)
    # heap layout (bytes): sha regions back to back from 0, then keccak regions (each preceded by 32 spare bytes)
    sha_off, off = [], 0
    for r in sha_rounds:
        sha_off.append(off)
        off += 64 * r
    kec_off = []
    for j, k in enumerate(keccak_k):
        off = (off + 31) // 32 * 32
        kec_off.append(off + kun[j])
        off += klen[j] + 32
    heap_words = (off + 31) // 32 + 8
    out_base_boot = heap_words - 6  # sha digests land in the last words of the bootloader heap
    boot_page_heap = BOOTLOADER_BASE_PAGE + 2
    consts = []
    for j, r in enumerate(sha_rounds):
        consts.append(precompile_abi(sha_off[j] // 32, 0, out_base_boot + j, 0, 0, 0, extra=r))
    consts.append(far_call_abi(isa, 0, heap_words * 32, 0x7FFFFFFF))  # [4] far-call ABI: whole heap as calldata
    consts.append(K.u256_from_int(KECCAK_ADDRESS))                # [5]
    ops = []
    for j, r in enumerate(sha_rounds):
        ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=CONST_BASE + j, src1=0, dst0=3))
        ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=SHA256_COST_PER_ROUND * r, src1=0, dst0=4))
        ops.append(isa.enc(K.OP_LOG, variant=K.LOG_PRECOMPILE, src0=3, src1=4, dst0=5 + j))
    ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=CONST_BASE + 4, src1=0, dst0=13))
    ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=CONST_BASE + 5, src1=0, dst0=14))
    ops.append(isa.enc(K.OP_FAR_CALL, variant=K.FAR_NORMAL, src0=13, src1=14, imm0=len(ops) + 1))
    n_boot_tail = 4
    for _ in range(n_boot_tail):
        ops.append(isa.enc(K.OP_UMA, variant=K.UMA_FAT_PTR_READ, flags=1, src0=1, dst0=3, dst1=1))  # read back the digests
    boot = np.zeros((CONST_BASE + len(consts), 4), dtype="<u8")
    code = K.pack_code(ops)
    boot[: len(code)] = code
    for i, c in enumerate(consts):
        boot[CONST_BASE + i] = c
    wl.blobs.append(boot)
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    # keccak system contract
    cops = []
    cconsts = []
    for j, k in enumerate(keccak_k):
        cconsts.append(precompile_abi(kec_off[j], klen[j], j, 0, boot_page_heap, 0))
    cconsts.append(ret_abi(isa, 0, 32 * len(keccak_k)))
    local = 64
    for j, k in enumerate(keccak_k):
        cops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local + j, src1=0, dst0=3))
        cost = KECCAK_COST_PER_ROUND * (k + 1)
        cops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local + 8 + j, src1=0, dst0=4))
        cconsts_cost = K.u256_from_int(cost)
        cops.append(isa.enc(K.OP_LOG, variant=K.LOG_PRECOMPILE, src0=3, src1=4, dst0=5 + j))
    cops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local + 4, src1=0, dst0=13))
    cops.append(isa.enc(K.OP_RET, variant=K.RET_OK, src0=13))
    callee = np.zeros((96, 4), dtype="<u8")
    code = K.pack_code(cops)
    callee[: len(code)] = code
    for i, c in enumerate(cconsts):
        callee[local + i] = c
    for j, k in enumerate(keccak_k):
        callee[local + 8 + j] = K.u256_from_int(KECCAK_COST_PER_ROUND * (k + 1))
    wl.blobs.append(callee)
    h = versioned_code_hash(callee)
    wl.preimages.append((h, 1))
    slots = np.zeros(1, dtype=K.STORAGE_SLOT)
    slots[0]["key"] = K.u256_from_int(KECCAK_ADDRESS)
    slots[0]["value"] = h
    slots[0]["address"] = K.address_bytes(0x8002)
    wl.storage = [slots] * n_instances
    executed = len(ops) + len(cops)
    assert executed <= n_cycles, (executed, n_cycles)
    wl.n_cycles = executed
    wl.limits["max_cycles"] = executed
    regs = np.zeros((n_instances, 15, 4), dtype="<u8")
    wl.states, wl.inner = initial_states(n_instances, regs, heap_bound=heap_words * 32 + 64)
    wl.states["current"]["this_address"] = K.address_bytes(SHA256_ADDRESS)
    wl.states["current"]["code_address"] = K.address_bytes(SHA256_ADDRESS)
    # per-instance random heap; sha regions hold properly padded messages so that the digests are real SHA-256 values
    heaps = Xoshiro(seed ^ 0x4EA9, n_instances).words(heap_words)
    raw = heaps.astype(">u8")[:, :, ::-1].copy()  # big-endian byte image [n, words, 4] -> bytes
    by = raw.view("u1").reshape(n_instances, heap_words * 32)
    wl.sha_messages = []
    for j, r in enumerate(sha_rounds):
        msg_len = 64 * r - 9 - (j * 5) % 40  # fits r blocks exactly
        start = sha_off[j]
        by[:, start + msg_len] = 0x80
        by[:, start + msg_len + 1: start + 64 * r - 8] = 0
        by[:, start + 64 * r - 8: start + 64 * r] = np.frombuffer((8 * msg_len).to_bytes(8, "big"), dtype="u1")
        wl.sha_messages.append((start, msg_len, out_base_boot + j))
    wl.keccak_messages = [(kec_off[j], klen[j], j) for j, k in enumerate(keccak_k)]
    wl.heap_bytes = by
    heaps = by.reshape(n_instances, heap_words, 4, 8).view(">u8").reshape(n_instances, heap_words, 4)[:, :, ::-1].astype("<u8")
    wl.heaps = np.ascontiguousarray(heaps)
    n_pre_reads = sum(2 * r for r in sha_rounds) + sum((b + 31 + 31) // 32 + 1 for b in klen)
    wl.limits.update(max_far_frames=2, heap_words=heap_words + 8, stack_words=8, aux_heap_words=8, storage_slots=8, storage_journal=4,
                     max_mem_queries=n_pre_reads + 8 * executed + 64, max_log_queries=16, max_aux_events=16)
    return wl


# ----------------------------------------------------------------------------------------
# cfg 4 — full synthetic L2 block: cfg-2 mix + storage r/w + events + L2->L1 + reverting near calls
# ----------------------------------------------------------------------------------------
N_STORAGE_KEYS = 256


class BlockTapeBuilder(TapeBuilder):
    def key_reg(self):
        # r11 := small storage key (pre-populated slots are keys 0..255)
        self.emit(self.isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=self.rng.below(N_STORAGE_KEYS + 32), src1=0, dst0=11))

    def dst(self):
        self.k += 1
        return 3 + (self.k % 8)  # r3..r10; r11 = storage key, r12 = heap cursor

    def src(self):
        return 1 + self.rng.below(10)

    def storage_read(self):
        self.key_reg()
        self.emit(self.isa.enc(K.OP_LOG, variant=K.LOG_STORAGE_READ, src0=11, src1=0, dst0=self.dst()))

    def storage_write(self):
        self.key_reg()
        self.emit(self.isa.enc(K.OP_LOG, variant=K.LOG_STORAGE_WRITE, src0=11, src1=self.src()))

    def event(self):
        self.emit(self.isa.enc(K.OP_LOG, variant=K.LOG_EVENT, flags=self.rng.below(2), src0=self.src(), src1=self.src()))

    def to_l1(self):
        self.emit(self.isa.enc(K.OP_LOG, variant=K.LOG_TO_L1, flags=self.rng.below(2), src0=self.src(), src1=self.src()))

    def context_op(self):
        r = self.rng
        v = r.below(10)
        if v <= K.CTX_GET_CONTEXT_U128:
            self.emit(self.isa.enc(K.OP_CONTEXT, variant=v, dst0=self.dst()))
        elif v == K.CTX_INC_TX_NUMBER:
            self.emit(self.isa.enc(K.OP_CONTEXT, variant=v))
        else:
            self.emit(self.isa.enc(K.OP_CONTEXT, variant=v, src0=self.src()))
            if v == K.CTX_SET_ERGS_PER_PUBDATA:  # keep pubdata prices small: immediately override with an immediate-derived value
                self.emit(self.isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=1 + r.below(20), src1=0, dst0=10))
                self.emit(self.isa.enc(K.OP_CONTEXT, variant=v, src0=10))

    def ptr_op(self):
        # r1 holds a fat pointer after any far-call return; results go to r2 (keeps the pointer tag)
        r = self.rng
        v = r.below(4)
        if v in (K.PTR_ADD, K.PTR_SHRINK):
            self.emit(self.isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=r.below(48), src1=0, dst0=10))
            self.emit(self.isa.enc(K.OP_PTR, variant=v, flags=0, src0=1, src1=10, dst0=2))
        elif v == K.PTR_SUB:  # advance first so that the subtraction cannot underflow (that would be a VM panic)
            self.emit(self.isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=48, src1=0, dst0=10))
            self.emit(self.isa.enc(K.OP_PTR, variant=K.PTR_ADD, flags=0, src0=1, src1=10, dst0=2))
            self.emit(self.isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=r.below(48), src1=0, dst0=10))
            if r.below(2):  # swapped form: the pointer travels in src1 and the swap flag puts it back into src0
                self.emit(self.isa.enc(K.OP_PTR, variant=K.PTR_SUB, flags=1, src0=10, src1=2, dst0=2))
            else:
                self.emit(self.isa.enc(K.OP_PTR, variant=K.PTR_SUB, flags=0, src0=2, src1=10, dst0=2))
        else:
            self.emit(self.isa.enc(K.OP_SHIFT, variant=K.SHIFT_SHL, src0_mode=K.MODE_REG, flags=0, src0=self.src(), src1=0, dst0=10))
            self.emit(self.isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=128, src1=0, dst0=9))
            self.emit(self.isa.enc(K.OP_SHIFT, variant=K.SHIFT_SHL, flags=0, src0=10, src1=9, dst0=10))  # low 128 bits cleared
            self.emit(self.isa.enc(K.OP_PTR, variant=K.PTR_PACK, flags=0, src0=1, src1=10, dst0=2))

    def block_segment(self, n):
        """cfg-2 mix + 8% storage read, 4% storage write, 3% event, 1% L2->L1 (+ context / ptr ops)."""
        target = self.executed + n
        self.set_cursor()
        while self.executed < target - 4:
            x = self.rng.below(100)
            if x < 8:
                self.storage_read()
            elif x < 12:
                self.storage_write()
            elif x < 15:
                self.event()
            elif x < 16:
                self.to_l1()
            elif x < 19:
                self.context_op()
            elif x < 22:
                self.ptr_op()
            elif x < 52:
                if self.rng.chance(1, 6):
                    self.set_cursor()
                else:
                    self.alu()
            elif x < 68:
                self.heap_ld()
            elif x < 84:
                self.heap_st()
            elif x < 92:
                self.stack_op()
            else:
                self.jump()
        while self.executed < target:
            self.emit(self.isa.enc(K.OP_NOP))


def config4(isa, n_instances=4096, n_cycles=1024, seed=0x5EED0004):
    wl = Workload("cfg4_l2_block", n_instances, n_cycles)
    rng = ScalarRng(seed)
    tb = BlockTapeBuilder(isa, rng)
    consts_rnd = Xoshiro(seed ^ 0x77, 1).words(4)[0]
    consts = [far_call_abi(isa, 64, 256, 100000), K.u256_from_int(ADDR_A), far_call_abi(isa, 512, 96, 50000), K.u256_from_int(ADDR_B),
              consts_rnd[0], consts_rnd[1], consts_rnd[2], consts_rnd[3]]
    # subroutines (near-call targets) are placed after the main program; their bodies write storage and
    # emit events so that a reverting return exercises the rollback journals
    SUB_BODY = 12
    n_near = max(1, n_cycles // 64)
    far_overhead = 2 * (1 + CALLEE_CYCLES + RELOAD_CYCLES)
    near_overhead = n_near * (2 + SUB_BODY + 1)
    main_budget = n_cycles - far_overhead - near_overhead
    assert main_budget > 64
    seg = main_budget // (n_near + 2)
    pending_subs = []  # (index of the near_call op in tb.ops, reverting?)
    far_at = {n_near // 3: (2, 3), (2 * n_near) // 3: (0, 1)}
    used = 0
    for j in range(n_near):
        tb.block_segment(seg)
        used += seg
        if j in far_at:
            tb.far_call()
            tb.executed += CALLEE_CYCLES
            tb.reload(*far_at[j])
        # near call: r10 := ergs to pass, then near_call r10, @sub, @handler
        tb.emit(isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=20000 + rng.below(20000), src1=0, dst0=10))
        pending_subs.append((len(tb.ops), rng.chance(1, 10) or j == 1))
        tb.emit(0)  # patched below
        tb.executed += SUB_BODY + 1
    tb.block_segment(n_cycles - tb.executed)
    assert tb.executed == n_cycles, (tb.executed, n_cycles)
    # jump-over guard, then the subroutines
    for idx, reverting in pending_subs:
        sub_pc = len(tb.ops)
        handler = idx + 1  # exception handler = fall through
        tb.ops[idx] = isa.enc(K.OP_NEAR_CALL, src0=10, imm0=sub_pc, imm1=handler)
        sub = BlockTapeBuilder(isa, rng)
        sub.ops = tb.ops
        for q in range(SUB_BODY // 4):
            sub.storage_write()   # 2 ops
            sub.event()           # 1 op
            sub.alu()             # 1 op
        tb.ops.append(isa.enc(K.OP_RET, variant=K.RET_REVERT if reverting else K.RET_OK, flags=0, src0=0))
    boot_words = np.zeros((CONST_BASE + len(consts), 4), dtype="<u8")
    code = K.pack_code(tb.ops)
    assert len(code) < CONST_BASE, len(code)
    boot_words[: len(code)] = code
    for i, c in enumerate(consts):
        boot_words[CONST_BASE + i] = c
    wl.blobs.append(boot_words)
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    for which, retv in ((0, K.RET_OK), (1, K.RET_REVERT)):
        ops = callee_program(isa, rng, retv)
        words = np.zeros((CALLEE_CODE_WORDS, 4), dtype="<u8")
        fill = Xoshiro(seed ^ (0x1000 + which), 1).words(CALLEE_CODE_WORDS - 64)[0]
        words[32: 32 + len(fill)] = fill
        words[CALLEE_CODE_WORDS - 8:] = 0
        local = CALLEE_CODE_WORDS - 8
        ops[11] = isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local, src1=0, dst0=13)
        code = K.pack_code(ops)
        words[: len(code)] = code
        words[local] = ret_abi(isa, 0, 64)
        wl.blobs.append(words)
        wl.preimages.append((versioned_code_hash(words), 1 + which))
    # storage: deployer entries + 256 pre-populated slots of the bootloader's own account
    n_slots = 2 + N_STORAGE_KEYS
    vals = Xoshiro(seed ^ 0x5107, n_instances).words(N_STORAGE_KEYS)
    wl.storage = []
    base = np.zeros(n_slots, dtype=K.STORAGE_SLOT)
    for which, addr in enumerate((ADDR_A, ADDR_B)):
        base[which]["key"] = K.u256_from_int(addr)
        base[which]["value"] = wl.preimages[which][0]
        base[which]["address"] = K.address_bytes(0x8002)
    for kx in range(N_STORAGE_KEYS):
        base[2 + kx]["key"] = K.u256_from_int(kx)
        base[2 + kx]["address"] = K.address_bytes(KERNEL_ADDRESS)
    for i in range(n_instances):
        s = base.copy()
        s["value"][2:] = vals[i]
        wl.storage.append(s)
    regs = Xoshiro(seed ^ 0xABCDEF, n_instances).words(15)
    regs[:, 5::4, 3] = 0
    regs[:, 6::4, 2:] = 0
    regs[:, 10] = 0
    regs[:, 11] = 0
    regs[:, 12] = consts[0]
    regs[:, 13] = consts[1]
    regs[:, 14] = 0
    wl.states, wl.inner = initial_states(n_instances, regs)
    wl.states["current_ergs_per_pubdata_byte"] = 7
    wl.states["register_ptr_bitmap"] = 1  # r1 starts as a (synthetic) fat pointer so that ptr.* ops before the first far call are legal
    wl.states["registers"][:, 0] = K.u256_from_int((0) | (0 << 32) | (0 << 64) | (4096 << 96))  # offset 0, page 0 (Empty), start 0, length 4096
    wl.heaps = Xoshiro(seed ^ 0x4EA9, n_instances).words(HEAP_BYTES // 32)
    n_writes = n_cycles  # generous
    wl.limits.update(max_far_frames=3, max_callstack_depth=6, heap_words=384, stack_words=256, aux_heap_words=8, storage_slots=1024,
                     storage_journal=n_writes, max_log_queries=n_cycles, max_aux_events=n_cycles // 2 + 64)
    return wl


# ----------------------------------------------------------------------------------------
# ecrecover precompile calls from a frame whose address is the ecrecover system contract (0x01)
# ----------------------------------------------------------------------------------------
ECRECOVER_ADDRESS = 0x01


def ecrecover_workload(isa, sig_words, tail_cycles=6):
    """sig_words: [n_instances][n_sigs][4] python ints, already in the memory order of isa.consts.ecrecover_input_layout.
    Signature j sits at heap words 4j..4j+3; the precompile writes (ok marker, address word) at words out_base + 2j,
    which the tail of the tape loads back into registers."""
    n_instances, n_sigs = len(sig_words), len(sig_words[0])
    out_base = 4 * n_sigs
    heap_words = out_base + 2 * n_sigs + 2
    ops, consts = [], []
    for j in range(n_sigs):
        consts.append(precompile_abi(4 * j, 4, out_base + 2 * j, 2, 0, 0))
        ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=CONST_BASE + j, src1=0, dst0=3))
        ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=1112, src1=0, dst0=4))
        ops.append(isa.enc(K.OP_LOG, variant=K.LOG_PRECOMPILE, src0=3, src1=4, dst0=5))
    for j in range(min(n_sigs, tail_cycles // 2)):  # read the results back through the heap
        ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=32 * (out_base + 2 * j + 1), src1=0, dst0=12))
        ops.append(isa.enc(K.OP_UMA, variant=K.UMA_HEAP_READ, flags=0, src0=12, dst0=6 + j))
    n_cycles = len(ops) + 2
    wl = Workload("ecrecover_%d" % n_sigs, n_instances, n_cycles)
    boot = np.zeros((CONST_BASE + len(consts), 4), dtype="<u8")
    code = K.pack_code(ops + [isa.enc(K.OP_NOP)] * 8)
    boot[: len(code)] = code
    for i, c in enumerate(consts):
        boot[CONST_BASE + i] = c
    wl.blobs.append(boot)
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    regs = np.zeros((n_instances, 15, 4), dtype="<u8")
    wl.states, wl.inner = initial_states(n_instances, regs, heap_bound=heap_words * 32 + 64)
    wl.states["current"]["this_address"] = K.address_bytes(ECRECOVER_ADDRESS)
    wl.states["current"]["code_address"] = K.address_bytes(ECRECOVER_ADDRESS)
    heaps = np.zeros((n_instances, heap_words, 4), dtype="<u8")
    for i in range(n_instances):
        for j in range(n_sigs):
            for k in range(4):
                heaps[i, 4 * j + k] = K.u256_from_int(sig_words[i][j][k])
    wl.heaps = heaps
    wl.out_base = out_base
    wl.limits.update(max_far_frames=2, heap_words=heap_words + 8, stack_words=8, aux_heap_words=8, storage_slots=8, storage_journal=4,
                     max_mem_queries=8 * n_sigs + 16, max_log_queries=16, max_aux_events=16)
    return wl


# ----------------------------------------------------------------------------------------
# nested near-call frames with storage writes / events / L1 messages and every ok / revert / panic
# combination — the frame discipline get_final_net_states depends on (testing/storage.rs:144-186,
# reference_impls/event_sink.rs:160-176)
# ----------------------------------------------------------------------------------------
def nested_frames(isa, outer=K.RET_OK, inner=K.RET_OK, main_panics=False, n_instances=3, n_cycles=40, seed=0x5EED00F2):
    wl = Workload("nested_frames_%d_%d_%d" % (outer, inner, int(main_panics)), n_instances, n_cycles)
    A, B = 16, 32
    e = isa.enc
    ops = [e(K.OP_NOP)] * 48
    def key(k):
        return e(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=k, src1=0, dst0=11)
    ops[0] = e(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=0, src1=0, dst0=10)  # near_call with 0 ergs = pass everything
    ops[1] = e(K.OP_NEAR_CALL, src0=10, imm0=A, imm1=2)
    ops[2] = e(K.OP_LOG, variant=K.LOG_EVENT, flags=1, src0=1, src1=2)
    ops[3] = key(5)
    ops[4] = e(K.OP_LOG, variant=K.LOG_STORAGE_WRITE, src0=11, src1=3)
    ops[5] = e(K.OP_LOG, variant=K.LOG_STORAGE_READ, src0=11, src1=0, dst0=4)
    ops[6] = e(K.OP_LOG, variant=K.LOG_TO_L1, flags=0, src0=4, src1=5)
    ops[7] = key(1)
    ops[8] = e(K.OP_LOG, variant=K.LOG_STORAGE_READ, src0=11, src1=0, dst0=6)
    ops[9] = e(K.OP_RET, variant=K.RET_PANIC) if main_panics else e(K.OP_JUMP, src0_mode=K.MODE_IMM, imm0=9)  # park
    # A: write key 1, event, call B, event, write key 1 again, write absent key 2, return
    ops[A + 0] = key(1)
    ops[A + 1] = e(K.OP_LOG, variant=K.LOG_STORAGE_WRITE, src0=11, src1=5)
    ops[A + 2] = e(K.OP_LOG, variant=K.LOG_EVENT, flags=0, src0=2, src1=3)
    ops[A + 3] = e(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=0, src1=0, dst0=10)
    ops[A + 4] = e(K.OP_NEAR_CALL, src0=10, imm0=B, imm1=A + 5)
    ops[A + 5] = e(K.OP_LOG, variant=K.LOG_EVENT, flags=0, src0=3, src1=4)
    ops[A + 6] = e(K.OP_LOG, variant=K.LOG_STORAGE_WRITE, src0=11, src1=6)
    ops[A + 7] = key(2)
    ops[A + 8] = e(K.OP_LOG, variant=K.LOG_STORAGE_WRITE, src0=11, src1=7)
    ops[A + 9] = e(K.OP_RET, variant=outer, flags=0, src0=0)
    # B: overwrite key 1, event, L1 message, return
    ops[B + 0] = key(1)
    ops[B + 1] = e(K.OP_LOG, variant=K.LOG_STORAGE_WRITE, src0=11, src1=8)
    ops[B + 2] = e(K.OP_LOG, variant=K.LOG_EVENT, flags=1, src0=5, src1=6)
    ops[B + 3] = e(K.OP_LOG, variant=K.LOG_TO_L1, flags=1, src0=6, src1=7)
    ops[B + 4] = e(K.OP_RET, variant=inner, flags=0, src0=0)
    wl.blobs.append(K.pack_code(ops))
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    vals = Xoshiro(seed ^ 0x5107, n_instances).words(2)
    wl.storage = []
    for i in range(n_instances):
        sl = np.zeros(2, dtype=K.STORAGE_SLOT)
        for j, k in enumerate((1, 5)):
            sl[j]["key"] = K.u256_from_int(k)
            sl[j]["value"] = vals[i][j]
            sl[j]["address"] = K.address_bytes(KERNEL_ADDRESS)
        wl.storage.append(sl)
    regs = Xoshiro(seed ^ 0xABCDEF, n_instances).words(15)
    wl.states, wl.inner = initial_states(n_instances, regs)
    wl.states["current_ergs_per_pubdata_byte"] = 3
    wl.limits.update(max_far_frames=2, max_callstack_depth=6, heap_words=8, stack_words=8, aux_heap_words=8, storage_slots=16, storage_journal=16,
                     max_log_queries=64, max_aux_events=32)
    return wl


# ----------------------------------------------------------------------------------------
# many sequential far calls under a small max_far_frames: the arena slot of a frame that has returned is reused unless
# its heap / aux heap became returndata (memory.rs:660-758).  Callees:
#   K  (kept)    writes its heap and returns a heap slice: the slot stays reachable from the caller (returndata)
#   P  (panics)  reads words of its fresh pages that the previous tenant of the slot had written (they must read zero),
#                writes heap + stack, then `ret.panic`: every page goes back to the pool
#   N  (nested)  far-calls K itself and return-forwards K's returndata pointer: K's page is handed to N's caller, N's own
#                slot goes back to the pool
# ----------------------------------------------------------------------------------------
def many_far_calls(isa, n_calls=64, n_instances=8, seed=0x5EED00F7, plan=None, max_far_frames=4, distinct=0):
    """distinct > 0: the plan calls `distinct` DIFFERENT contracts (copies of P with their own code hash and address, "P0" ..)
    one after the other and then the first few again (decommits that are no longer fresh): SimpleDecommitter's history is
    unbounded (decommitter.rs:38-47) — the decommits of a run are not capped by limits.max_far_frames"""
    e = isa.enc
    ADDR_K, ADDR_P, ADDR_N = 0x10011, 0x10012, 0x10013
    local = CALLEE_CODE_WORDS - 8  # page-local constant pool of the callees

    def callee_k():
        return [e(K.OP_UMA, variant=K.UMA_FAT_PTR_READ, flags=1, src0=1, dst0=3, dst1=1),
                e(K.OP_UMA, variant=K.UMA_HEAP_READ, src0_mode=K.MODE_IMM, imm0=96, dst0=4),          # fresh heap: zero
                e(K.OP_ADD, src0_mode=K.MODE_STACK_ABS, imm0=7, src1=0, dst0=5),                      # fresh stack: zero
                e(K.OP_ADD, flags=0, src0=3, src1=4, dst0=6),
                e(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=0, src1=6),
                e(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=96, src1=3),
                e(K.OP_ADD, dst0_mode=K.MODE_STACK_ABS, src0=3, src1=5, dst0=0, imm1=7),             # stack[7] := ...
                e(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local, src1=0, dst0=13),
                e(K.OP_RET, variant=K.RET_OK, flags=0, src0=13)]

    def callee_p():
        return [e(K.OP_UMA, variant=K.UMA_HEAP_READ, src0_mode=K.MODE_IMM, imm0=0, dst0=3),
                e(K.OP_UMA, variant=K.UMA_HEAP_READ, src0_mode=K.MODE_IMM, imm0=96, dst0=4),
                e(K.OP_UMA, variant=K.UMA_AUX_READ, src0_mode=K.MODE_IMM, imm0=32, dst0=5),
                e(K.OP_ADD, src0_mode=K.MODE_STACK_ABS, imm0=7, src1=0, dst0=6),
                e(K.OP_UMA, variant=K.UMA_FAT_PTR_READ, flags=0, src0=1, dst0=7),
                e(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=0, src1=7),
                e(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=100, src1=7),
                e(K.OP_ADD, dst0_mode=K.MODE_STACK_ABS, src0=7, src1=3, dst0=0, imm1=7),
                e(K.OP_RET, variant=K.RET_PANIC, flags=0, src0=0)]

    def callee_n():
        return [e(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local + 1, src1=0, dst0=13),   # far-call ABI
                e(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local + 2, src1=0, dst0=14),   # callee K
                e(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=32, src1=13),
                e(K.OP_FAR_CALL, variant=K.FAR_NORMAL, src0=13, src1=14, imm0=4),
                e(K.OP_UMA, variant=K.UMA_FAT_PTR_READ, flags=0, src0=1, dst0=5),      # K's returndata
                e(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local + 3, src1=0, dst0=12),
                e(K.OP_PTR, variant=K.PTR_PACK, src0=1, src1=12, dst0=13),            # returndata pointer | forwarding mode 1 in the top bits
                e(K.OP_RET, variant=K.RET_OK, flags=0, src0=13)]

    programs = {"K": (ADDR_K, callee_k()), "P": (ADDR_P, callee_p()), "N": (ADDR_N, callee_n())}
    order = ["K", "P", "N"]
    if distinct:
        for j in range(distinct):
            programs["P%d" % j] = (0x10100 + j, callee_p())
            order.append("P%d" % j)
        again = min(8, distinct)
        plan = ["K"] + ["P%d" % j for j in range(distinct)] + ["P%d" % j for j in range(again)] + ["N"] + ["P%d" % (distinct - 1 - j) for j in range(again)]
        n_calls = len(plan)
    if plan is None:
        plan = "K" + "P" * 5 + "N" + "P" * (n_calls - 7)  # two live returndata pages + the bootloader: one slot left to reuse
    assert len(plan) == n_calls
    n_cycles = 4
    for c in plan:
        n_cycles += 1 + len(programs[c][1]) + (len(programs["K"][1]) if c == "N" else 0) + RELOAD_CYCLES
    wl = Workload("many_far_calls_%s" % "".join(plan)[:12], n_instances, n_cycles)
    rng = ScalarRng(seed)
    tb = TapeBuilder(isa, rng)
    consts = []
    for c in order:
        consts += [far_call_abi(isa, 64, 128, 200000), K.u256_from_int(programs[c][0])]
    for j, c in enumerate(plan):  # r13 / r14 hold the ABI / address of callee j: preset for the first, reloaded after every return
        tb.far_call()
        k = order.index(plan[min(j + 1, n_calls - 1)])
        tb.reload(2 * k, 2 * k + 1)   # (also reads the first returndata word through r1)
    for _ in range(4):
        tb.emit(e(K.OP_ADD, flags=1, src0=12, src1=2, dst0=3))
    boot_words = np.zeros((CONST_BASE + len(consts), 4), dtype="<u8")
    code = K.pack_code(tb.ops)
    assert len(code) < CONST_BASE
    boot_words[: len(code)] = code
    for i, c in enumerate(consts):
        boot_words[CONST_BASE + i] = c
    wl.blobs.append(boot_words)
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    slots = np.zeros(len(order), dtype=K.STORAGE_SLOT)
    for which, c in enumerate(order):
        addr, ops = programs[c]
        words = np.zeros((CALLEE_CODE_WORDS, 4), dtype="<u8")
        code = K.pack_code(ops)
        words[: len(code)] = code
        fill = Xoshiro(seed ^ (0x2000 + which), 1).words(64)[0]
        words[64:128] = fill
        words[local] = ret_abi(isa, 0, 128)
        words[local + 1] = far_call_abi(isa, 0, 64, 50000)
        words[local + 2] = K.u256_from_int(ADDR_K)
        words[local + 3] = K.u256_from_int(isa.fwd_code(1) << 224)  # RetABI forwarding_mode = ForwardFatPointer, in the half ptr.pack takes from src1
        wl.blobs.append(words)
        h = versioned_code_hash(words)
        wl.preimages.append((h, 1 + which))
        slots[which]["key"] = K.u256_from_int(addr)
        slots[which]["value"] = h
        slots[which]["address"] = K.address_bytes(0x8002)
    wl.storage = [slots] * n_instances
    regs = Xoshiro(seed ^ 0xABCDEF, n_instances).words(15)
    regs[:, 11] = 0
    regs[:, 12] = consts[2 * order.index(plan[0])]
    regs[:, 13] = consts[2 * order.index(plan[0]) + 1]
    regs[:, 14] = 0
    wl.states, wl.inner = initial_states(n_instances, regs)
    wl.heaps = Xoshiro(seed ^ 0x4EA9, n_instances).words(HEAP_BYTES // 32)
    wl.limits.update(max_far_frames=max_far_frames, heap_words=320, stack_words=16, aux_heap_words=8, storage_slots=max(8, 4 * len(order)), storage_journal=4,
                     max_aux_events=8 * n_calls + 32, max_reg_deltas=2 * n_cycles + 40 * n_calls)
    return wl



# ----------------------------------------------------------------------------------------
# the bootloader itself returns (the instance ENDS) with its heap / aux heap as returndata, or panics: what
# `dump_page_content` sees of the bootloader's pages afterwards follows finish_global_frame (memory.rs:660-758) — the stack
# page and the page that is not the returndata go back to the pool
# ----------------------------------------------------------------------------------------
def bootloader_returns(isa, how="heap", n_instances=3, seed=0x5EED00F9):
    e = isa.enc
    wl = Workload("bootloader_returns_%s" % how, n_instances, 8)
    ops = [e(K.OP_ADD, dst0_mode=K.MODE_STACK_ABS, src0=1, src1=2, dst0=0, imm1=7),               # stack[7]
           e(K.OP_UMA, variant=K.UMA_AUX_WRITE, src0_mode=K.MODE_IMM, imm0=0, src1=3),            # aux[0]
           e(K.OP_UMA, variant=K.UMA_HEAP_WRITE, src0_mode=K.MODE_IMM, imm0=40, src1=4),          # heap[1..2]
           e(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=16, src1=0, dst0=13),
           e(K.OP_RET, variant=K.RET_PANIC if how == "panic" else K.RET_OK, flags=0, src0=13)]
    words = np.zeros((24, 4), dtype="<u8")
    code = K.pack_code(ops)
    words[: len(code)] = code
    words[16] = ret_abi(isa, 0, 128, forwarding_mode={"heap": 0, "aux": 2, "panic": 0}[how])
    wl.blobs.append(words)
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    regs = Xoshiro(seed ^ 0xABCDEF, n_instances).words(15)
    wl.states, wl.inner = initial_states(n_instances, regs)
    wl.heaps = Xoshiro(seed ^ 0x4EA9, n_instances).words(16)
    wl.limits.update(max_far_frames=2, heap_words=32, stack_words=16, aux_heap_words=8)
    return wl


def make(cfg, isa, **kw):  # noqa: F811
    return {0: config0, 1: config1, 2: config2, 3: config3, 4: config4}[cfg](isa, **kw)


# ----------------------------------------------------------------------------------------
# fuzz tapes: every instance runs its own tape of random VALID encodings (any opcode variant, any operand mode,
# condition and register numbers, small / page-sized / arbitrary immediates).  Used by the parity tests to drive the
# rarely used paths (pointer arithmetic, context getters / setters, events / L1 messages, near calls, returns of
# every kind, stack addressing modes, failing far calls) through the kernel and the oracle on the same inputs.
# Precompile calls are left to cfg 3 (a random ABI would ask for up to 2^32 rounds).
# ----------------------------------------------------------------------------------------
def _shaped_u256(rng):
    bits = (0, 5, 8, 13, 16, 32, 64, 128, 256)[rng.below(9)]
    if bits == 0:
        return 0
    v = 0
    for _ in range(4):
        v = (v << 64) | rng.u64()
    return v & ((1 << bits) - 1)


def fuzz_workload(isa, n_instances=64, n_ops=96, seed=0xF022):
    wl = Workload("fuzz_%x" % seed, n_instances, n_ops)
    rng = ScalarRng(seed)
    e = isa.table["entries"][0]
    classes = {}
    for idx in range(len(e)):
        op, var = int(e["opcode"][idx]), int(e["variant"][idx])
        if op == K.OP_LOG and var == K.LOG_PRECOMPILE:
            continue
        classes.setdefault(op, []).append(idx)
    weights = {K.OP_INVALID: 1, K.OP_RET: 2, K.OP_FAR_CALL: 3, K.OP_NEAR_CALL: 4, K.OP_JUMP: 4}
    wheel = []
    for op, lst in sorted(classes.items()):
        wheel += [op] * weights.get(op, 8)
    # the cfg-2 callees, reachable through DEPLOYER[ADDR_A / ADDR_B]
    callee_rng = ScalarRng(seed ^ 0xCA11)
    callee_words = []
    for which, retv in ((0, K.RET_OK), (1, K.RET_REVERT)):
        ops = callee_program(isa, callee_rng, retv)
        local = CALLEE_CODE_WORDS - 8
        ops[11] = isa.enc(K.OP_ADD, src0_mode=K.MODE_CODE, imm0=local, src1=0, dst0=13)
        words = np.zeros((CALLEE_CODE_WORDS, 4), dtype="<u8")
        code = K.pack_code(ops)
        words[: len(code)] = code
        words[local] = ret_abi(isa, 0, 64)
        callee_words.append(words)
    n_words = (n_ops + 3) // 4 + 8
    regs = np.zeros((n_instances, 15, 4), dtype="<u8")
    seg = 16  # ops per segment: ergs, near_call (handler = next segment), jump to the next segment, 12 random ops, ret.ok
    for i in range(n_instances):
        tape = []
        for pc in range(n_ops):
            base = pc - pc % seg
            nxt = min(n_ops - 1, base + seg)
            if pc % seg == 0:    # ergs for the segment (a panic forfeits what its frame was given)
                tape.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=20000, src1=0, dst0=15))
                continue
            if pc % seg == 1:    # a panic inside the segment unwinds to the next one instead of ending the instance
                tape.append(isa.enc(K.OP_NEAR_CALL, src0=15, imm0=min(n_ops - 1, pc + 2), imm1=nxt))
                continue
            if pc % seg == 2:
                tape.append(isa.enc(K.OP_JUMP, src0_mode=K.MODE_IMM, imm0=nxt))
                continue
            if pc % seg == seg - 1:
                tape.append(isa.enc(K.OP_RET, variant=K.RET_OK, src0=rng.below(16)))
                continue
            op = wheel[rng.below(len(wheel))]
            lst = classes[op]
            idx = lst[rng.below(len(lst))]
            cond = K.COND_ALWAYS if rng.below(10) < 7 else rng.below(8)
            r = [rng.below(16) for _ in range(4)]
            imms = []
            for _ in range(2):
                x = rng.below(4)
                imms.append(rng.below(64) if x < 2 else (32 * rng.below(64) if x == 2 else rng.below(1 << 16)))
            if op in (K.OP_JUMP, K.OP_NEAR_CALL) and rng.below(4) != 0:  # mostly forward, inside the segment
                imms[0] = min(base + seg - 1, pc + 1 + rng.below(6))
                imms[1] = min(base + seg - 1, pc + 1 + rng.below(12))
            if op == K.OP_FAR_CALL and rng.below(2) == 0:
                r[0], r[1] = 13, 14
                imms[0] = min(n_ops - 1, pc + 1)
            tape.append((idx & 0x7FF) | (cond << 13) | (r[0] << 16) | (r[1] << 20) | (r[2] << 24) | (r[3] << 28) | (imms[0] << 32) | (imms[1] << 48))
        words = np.zeros((n_words, 4), dtype="<u8")
        code = K.pack_code(tape)
        words[: len(code)] = code
        for k in range(len(code), n_words):  # constants for MODE_CODE operands that land behind the tape
            words[k] = K.u256_from_int(_shaped_u256(rng))
        wl.blobs.append(words)
        wl.code_pages.append((i, 1, BOOTLOADER_CODE_PAGE, i))
        for k in range(15):
            regs[i, k] = K.u256_from_int(_shaped_u256(rng))
        regs[i, 12] = far_call_abi(isa, 32 * rng.below(8), 32 * rng.below(8), 20000 + rng.below(1 << 20))  # r13
        regs[i, 13] = K.u256_from_int((ADDR_A, ADDR_B)[rng.below(2)])                                 # r14
    for which, words in enumerate(callee_words):
        wl.blobs.append(words)
        wl.preimages.append((versioned_code_hash(words), n_instances + which))
    slots = np.zeros(2, dtype=K.STORAGE_SLOT)
    for which, addr in enumerate((ADDR_A, ADDR_B)):
        slots[which]["key"] = K.u256_from_int(addr)
        slots[which]["value"] = wl.preimages[which][0]
        slots[which]["address"] = K.address_bytes(0x8002)
        slots[which]["shard_id"] = 0
    wl.storage = [slots] * n_instances
    # capacities that random operands cannot overrun: the whole 2^16-word stack page, and heaps as large as the ergs
    # can pay for (growth costs one erg per byte, uma.rs:196-207)
    ergs = 1 << 20
    wl.states, wl.inner = initial_states(n_instances, regs, ergs=ergs)
    wl.heaps = Xoshiro(seed ^ 0x4EA9, n_instances).words(64)
    wl.limits.update(max_far_frames=8, max_callstack_depth=32, heap_words=(ergs + 4096) // 32 + 2, stack_words=1 << 16, aux_heap_words=(ergs + 4096) // 32 + 2,
                     storage_slots=64, storage_journal=64)
    return wl


def uniform_fuzz(isa, n_instances=128, n_ops=192, seed=0xF100):
    """ONE random tape for every instance, per-instance registers and heaps: the lanes of a wave stay at one pc (what the cycle
    kernel's short cycle needs) while everything data-dependent differs between them — condition flags (a conditional
    instruction runs in some lanes and is a nop in others), operand values, heap offsets and alignments held in registers.
    Light opcodes with register / immediate / code-page operands and heap / aux-heap accesses dominate; stack operands, a
    fat-pointer-free mix of context reads and a few forward jumps ride along.  r1..r4 are address registers (small values:
    only cursor set-ups and post-increments write them), so that an access rarely leaves the heap and a panic stays rare."""
    wl = Workload("uniform_fuzz_%x" % seed, n_instances, n_ops)
    rng = ScalarRng(seed)
    ops = []
    n_consts = 8

    def cond():
        return K.COND_ALWAYS if rng.below(10) < 6 else rng.below(8)

    def src():
        return rng.below(16)

    def dst():
        return 0 if rng.below(12) == 0 else 5 + rng.below(11)

    def addr():
        return 1 + rng.below(4)

    while len(ops) < n_ops:
        x = rng.below(100)
        if x < 32:  # ALU, register / immediate / code-page operands
            which = rng.below(7)
            mode = (K.MODE_REG, K.MODE_REG, K.MODE_IMM, K.MODE_CODE)[rng.below(4)]
            kw = dict(src0_mode=mode, cond=cond(), src0=src() if mode == K.MODE_REG else 0, src1=src(), dst0=dst(),
                      imm0=(CONST_BASE + rng.below(n_consts + 2)) if mode == K.MODE_CODE else rng.below(1 << 16))
            sf = rng.below(2)
            if which == 0:
                ops.append(isa.enc(K.OP_ADD, flags=sf, **kw))
            elif which == 1:
                ops.append(isa.enc(K.OP_SUB, flags=sf | (rng.below(2) << 1), **kw))
            elif which == 2:
                ops.append(isa.enc(K.OP_MUL, flags=sf, dst1=dst(), **kw))
            elif which == 3:
                ops.append(isa.enc(K.OP_DIV, flags=sf | (rng.below(2) << 1), dst1=dst(), **kw))
            elif which == 4:
                ops.append(isa.enc(K.OP_SHIFT, variant=rng.below(4), flags=sf | (rng.below(2) << 1), **kw))
            else:
                ops.append(isa.enc(K.OP_BINOP, variant=rng.below(3), flags=sf, **kw))
        elif x < 70:  # heap / aux-heap accesses: immediate or register offsets, with and without post-increment
            variant = (K.UMA_HEAP_READ, K.UMA_HEAP_WRITE, K.UMA_HEAP_READ, K.UMA_HEAP_WRITE, K.UMA_AUX_READ, K.UMA_AUX_WRITE)[rng.below(6)]
            write = variant in (K.UMA_HEAP_WRITE, K.UMA_AUX_WRITE)
            if rng.below(2):
                a = addr()
                inc = rng.below(2)
                if write:
                    ops.append(isa.enc(K.OP_UMA, variant=variant, src0_mode=K.MODE_REG, flags=inc, cond=cond(), src0=a, src1=src(), dst0=a if rng.below(4) else addr()))
                else:
                    d = dst()
                    ops.append(isa.enc(K.OP_UMA, variant=variant, src0_mode=K.MODE_REG, flags=inc, cond=cond(), src0=a, dst0=d, dst1=a if rng.below(4) else addr()))
            else:
                y = rng.below(8)
                off = 32 * rng.below(100) if y < 3 else (rng.below(3200) if y < 7 else 4000 + rng.below(300))  # (the last: across the bound paid for)
                ops.append(isa.enc(K.OP_UMA, variant=variant, src0_mode=K.MODE_IMM, flags=0, cond=cond(), imm0=off, src1=src(), dst0=0 if write else dst()))
        elif x < 80:  # cursor set-ups: a small per-lane value into an address register
            a = addr()
            ops.append(isa.enc(K.OP_BINOP, variant=K.BINOP_AND, src0_mode=K.MODE_IMM, imm0=(31, 63, 255, 1023)[rng.below(4)], src1=5 + rng.below(11), dst0=a))
            if len(ops) < n_ops and rng.below(2):
                ops.append(isa.enc(K.OP_ADD, src0_mode=K.MODE_IMM, imm0=32 * rng.below(64), src1=a, dst0=a))
        elif x < 86:  # stack operands (the general path)
            if rng.below(2):
                ops.append(isa.enc(K.OP_ADD, dst0_mode=K.MODE_STACK_PP, cond=cond(), src0=src(), src1=src(), dst0=0, imm1=1))
            else:
                ops.append(isa.enc(K.OP_BINOP, variant=K.BINOP_XOR, src0_mode=K.MODE_STACK_OFF, src0=0, imm0=1 + rng.below(2), src1=src(), dst0=dst()))
        elif x < 91:  # context reads
            ops.append(isa.enc(K.OP_CONTEXT, variant=(K.CTX_THIS, K.CTX_CALLER, K.CTX_ERGS_LEFT, K.CTX_SP, K.CTX_META)[rng.below(5)], cond=cond(), dst0=dst()))
        elif x < 95:
            ops.append(isa.enc(K.OP_NOP, cond=cond()))
        else:  # a forward jump over up to three instructions; one in three conditional (the lanes then part for a few cycles and meet again)
            skip = rng.below(4)
            target = len(ops) + 1 + skip
            ops.append(isa.enc(K.OP_JUMP, src0_mode=K.MODE_IMM, imm0=target, cond=K.COND_ALWAYS if rng.below(3) else rng.below(8)))
            for _ in range(skip):
                ops.append(isa.enc(K.OP_ADD, src0=src(), src1=src(), dst0=dst()))
    ops = ops[:n_ops] + [isa.enc(K.OP_NOP)] * 8
    code = K.pack_code(ops)
    assert len(code) < CONST_BASE
    words = np.zeros((CONST_BASE + n_consts, 4), dtype="<u8")
    words[: len(code)] = code
    for k in range(n_consts):
        words[CONST_BASE + k] = K.u256_from_int(_shaped_u256(rng))
    wl.blobs.append(words)
    wl.code_pages.append((0, n_instances, BOOTLOADER_CODE_PAGE, 0))
    regs = np.zeros((n_instances, 15, 4), dtype="<u8")
    for i in range(n_instances):
        for k in range(15):
            regs[i, k] = K.u256_from_int(_shaped_u256(rng))
        for k in range(4):  # r1..r4: offsets inside the heap, every alignment
            regs[i, k] = K.u256_from_int(rng.below(3000))
    wl.states, wl.inner = initial_states(n_instances, regs, ergs=1 << 24)
    wl.heaps = Xoshiro(seed ^ 0x4EA9, n_instances).words(HEAP_BYTES // 32)
    wl.limits.update(max_far_frames=2, heap_words=320, stack_words=256, aux_heap_words=320, storage_slots=8, storage_journal=4)
    return wl
