"""ctypes / numpy binding of include/zkw.h.

`Backend` wraps one shared library and one symbol prefix: the product is libzkw.so, prefix `zkw_`
(HIP kernels; needs a GPU to run).  (The tests drive their CPU checker through the same class with
another library and prefix — tests/_oracle.py; nothing in this package refers to it.)
Structs are mirrored as numpy structured dtypes (byte-exact with the C layout; checked by
tests/test_capi_layout.py through zkw_abi_sizeof).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

# ----------------------------------------------------------------------------------------
# constants of zkw.h
# ----------------------------------------------------------------------------------------
OK = 0
ERR_INVALID, ERR_DEVICE, ERR_LIMIT, ERR_NOT_RUN = -1, -2, -3, -4
STATUS_RUNNING, STATUS_ENDED, STATUS_UNKNOWN_CODE_HASH, STATUS_REFERENCE_PANIC, STATUS_LIMIT = 0, 1, 2, 3, 4
(OP_INVALID, OP_NOP, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_JUMP, OP_CONTEXT, OP_SHIFT, OP_BINOP, OP_PTR, OP_NEAR_CALL, OP_LOG, OP_FAR_CALL,
 OP_RET, OP_UMA) = range(16)
(CTX_THIS, CTX_CALLER, CTX_CODE_ADDRESS, CTX_META, CTX_ERGS_LEFT, CTX_SP, CTX_GET_CONTEXT_U128, CTX_SET_CONTEXT_U128,
 CTX_SET_ERGS_PER_PUBDATA, CTX_INC_TX_NUMBER) = range(10)
SHIFT_SHL, SHIFT_SHR, SHIFT_ROL, SHIFT_ROR = range(4)
BINOP_XOR, BINOP_AND, BINOP_OR = range(3)
PTR_ADD, PTR_SUB, PTR_PACK, PTR_SHRINK = range(4)
LOG_STORAGE_READ, LOG_STORAGE_WRITE, LOG_TO_L1, LOG_EVENT, LOG_PRECOMPILE = range(5)
FAR_NORMAL, FAR_DELEGATE, FAR_MIMIC = range(3)
RET_OK, RET_REVERT, RET_PANIC = range(3)
UMA_HEAP_READ, UMA_HEAP_WRITE, UMA_AUX_READ, UMA_AUX_WRITE, UMA_FAT_PTR_READ = range(5)
MODE_REG, MODE_STACK_PP, MODE_STACK_OFF, MODE_STACK_ABS, MODE_IMM, MODE_CODE = range(6)
PROP_EXPLICIT_PANIC, PROP_KERNEL_ONLY, PROP_STATIC_OK, PROP_SWAP, PROP_SRC0_PTR_OK, PROP_SRC1_PTR_OK = 1, 2, 4, 8, 16, 32
COND_ALWAYS, COND_GT, COND_LT, COND_EQ, COND_GE, COND_LE, COND_NE, COND_GT_OR_LT = range(8)
MEM_STACK, MEM_CODE, MEM_HEAP, MEM_AUX_HEAP, MEM_FAT_PTR = range(5)
MQ_TYPE_MASK, MQ_IS_PTR, MQ_RW, MQ_KIND_SHIFT = 0x07, 0x08, 0x10, 5
LQ_LOG, LQ_REFUND = 0, 1
LQ_RW, LQ_ROLLBACK, LQ_IS_SERVICE = 1, 2, 4
AUX_FRAME_START, AUX_FRAME_FINISH, AUX_DECOMMIT, AUX_COLD_STATE = 1, 2, 3, 4
ISA_TABLE_SIZE = 2048
QUEUE_MEMORY, QUEUE_LOG, QUEUE_DECOMMIT, QUEUE_COUNT = 0, 1, 2, 3

# ----------------------------------------------------------------------------------------
# numpy dtypes (explicit offsets, itemsize = C sizeof)
# ----------------------------------------------------------------------------------------
U256 = np.dtype(("<u8", (4,)))


def _dt(fields, itemsize):
    names, formats, offsets = zip(*fields)
    return np.dtype({"names": list(names), "formats": list(formats), "offsets": list(offsets), "itemsize": itemsize})


ISA_ENTRY = _dt([("opcode", "u1", 0), ("variant", "u1", 1), ("src0_mode", "u1", 2), ("dst0_mode", "u1", 3), ("flags", "u1", 4), ("props", "u1", 5),
                 ("reserved", "<u2", 6), ("price", "<u4", 8)], 12)
ISA_CONSTS = _dt([("nop_encoding", "<u8", 0), ("exception_revert_encoding", "<u8", 8), ("panic_variant_idx", "<u4", 16), ("nop_variant_idx", "<u4", 20),
                  ("clip_mode", "<u4", 24), ("time_delta_per_cycle", "<u4", 28), ("new_memory_pages_per_far_call", "<u4", 32),
                  ("vm_max_stack_depth", "<u4", 36), ("initial_sp_on_far_call", "<u4", 40), ("new_frame_memory_stipend", "<u4", 44),
                  ("memory_growth_ergs_per_byte", "<u4", 48), ("ergs_per_code_word_decommittment", "<u4", 52),
                  ("initial_storage_write_pubdata_bytes", "<u4", 56), ("l1_message_pubdata_bytes", "<u4", 60), ("max_offset_to_deref_low", "<u4", 64),
                  ("deployer_address_low", "<u4", 68), ("keccak_precompile_address", "<u4", 72), ("sha256_precompile_address", "<u4", 76),
                  ("ecrecover_precompile_address", "<u4", 80), ("storage_aux_byte", "u1", 84), ("event_aux_byte", "u1", 85),
                  ("l1_message_aux_byte", "u1", 86), ("precompile_aux_byte", "u1", 87), ("ecrecover_input_layout", "<u4", 88), ("bootloader_calldata_page", "<u4", 92),
                  ("call_regs", "<u4", 96), ("call_ranges", "<u4", 100), ("ret_regs", "<u4", 104), ("forwarding_codes", "<u4", 108), ("unmapped_page", "<u4", 112),
                  ("reserved0", "<u4", 116), ("max_offset_for_add_sub", "<u8", 120), ("condition_lut", "<u8", 128)], 136)
ISA_TABLE = _dt([("entries", (ISA_ENTRY, ISA_TABLE_SIZE), 0), ("consts", ISA_CONSTS, 12 * ISA_TABLE_SIZE)], 12 * ISA_TABLE_SIZE + 136)

CALLSTACK_ENTRY = _dt([("this_address", ("u1", 20), 0), ("msg_sender", ("u1", 20), 20), ("code_address", ("u1", 20), 40), ("base_memory_page", "<u4", 60),
                       ("code_page", "<u4", 64), ("sp", "<u2", 68), ("pc", "<u2", 70), ("exception_handler_location", "<u2", 72), ("is_static", "u1", 74),
                       ("is_local_frame", "u1", 75), ("ergs_remaining", "<u4", 76), ("this_shard_id", "u1", 80), ("caller_shard_id", "u1", 81),
                       ("code_shard_id", "u1", 82), ("reserved0", "u1", 83), ("reserved1", "<u4", 84), ("context_u128_value", ("<u8", 2), 88),
                       ("heap_bound", "<u4", 104), ("aux_heap_bound", "<u4", 108)], 112)
VM_LOCAL_STATE = _dt([("previous_code_word", U256, 0), ("registers", (U256, 15), 32), ("register_ptr_bitmap", "<u2", 512), ("flags", "u1", 514),
                      ("pending_exception", "u1", 515), ("previous_code_memory_page", "<u4", 516), ("timestamp", "<u4", 520),
                      ("monotonic_cycle_counter", "<u4", 524), ("spent_pubdata_counter", "<u4", 528), ("memory_page_counter", "<u4", 532),
                      ("absolute_execution_step", "<u4", 536), ("current_ergs_per_pubdata_byte", "<u4", 540), ("tx_number_in_block", "<u2", 544),
                      ("previous_super_pc", "<u2", 546), ("callstack_depth", "<u4", 548), ("context_u128_register", ("<u8", 2), 552),
                      ("current", CALLSTACK_ENTRY, 568)], 680)
BLOCK_PROPERTIES = _dt([("default_aa_code_hash", U256, 0), ("zkporter_is_available", "<u4", 32), ("reserved0", "<u4", 36)], 40)
STORAGE_SLOT = _dt([("key", U256, 0), ("value", U256, 32), ("address", ("u1", 20), 64), ("shard_id", "u1", 84), ("reserved0", ("u1", 3), 85)], 88)
LIMITS = _dt([("max_cycles", "<u4", 0), ("max_far_frames", "<u4", 4), ("max_callstack_depth", "<u4", 8), ("stack_words", "<u4", 12), ("heap_words", "<u4", 16),
              ("aux_heap_words", "<u4", 20), ("storage_slots", "<u4", 24), ("storage_journal", "<u4", 28), ("max_mem_queries", "<u4", 32),
              ("max_log_queries", "<u4", 36), ("max_aux_events", "<u4", 40), ("lanes_per_wave", "<u4", 44), ("max_reg_deltas", "<u4", 48), ("reserved", ("<u4", 3), 52)], 64)
CYCLE_TAIL = _dt([("register_ptr_bitmap", "<u2", 0), ("flags", "u1", 2), ("reserved0", "u1", 3), ("pc", "<u2", 4), ("sp", "<u2", 6), ("ergs_remaining", "<u4", 8),
                  ("timestamp", "<u4", 12), ("heap_bound", "<u4", 16), ("aux_heap_bound", "<u4", 20), ("callstack_depth", "<u2", 24),
                  ("previous_super_pc", "<u2", 26), ("event_counts", "<u4", 28)], 32)
CYCLE_RECORD = _dt([("registers", (U256, 15), 0), ("tail", CYCLE_TAIL, 480)], 512)
MEM_QUERY = _dt([("timestamp", "<u4", 0), ("page", "<u4", 4), ("index", "<u4", 8), ("lane", "u1", 12), ("seq", "u1", 13), ("meta", "u1", 14), ("reserved0", "u1", 15),
                 ("value", U256, 16)], 48)
LOG_QUERY = _dt([("key", U256, 0), ("read_value", U256, 32), ("written_value", U256, 64), ("address", ("u1", 20), 96), ("timestamp", "<u4", 116),
                 ("tx_number_in_block", "<u2", 120), ("aux_byte", "u1", 122), ("shard_id", "u1", 123), ("bools", "u1", 124), ("kind", "u1", 125),
                 ("lane", "u1", 126), ("seq", "u1", 127)], 128)
AUX_EVENT = _dt([("type", "u1", 0), ("lane", "u1", 1), ("seq", "u1", 2), ("flag", "u1", 3), ("a", "<u4", 4), ("b", "<u4", 8), ("c", "<u4", 12),
                 ("raw", ("u1", 240), 16)], 256)
RUN_STATS = _dt([("cycles", "<u8", 0), ("mem_queries", "<u8", 8), ("log_queries", "<u8", 16), ("aux_events", "<u8", 24), ("instances_ended", "<u8", 32),
                 ("instances_failed", "<u8", 40), ("kernel_ms", "<f8", 48), ("reg_deltas", "<u8", 56)], 64)


EVENT_MESSAGE = _dt([("shard_id", "u1", 0), ("is_first", "u1", 1), ("tx_number_in_block", "<u2", 2), ("address", ("u1", 20), 4), ("key", U256, 24),
                     ("value", U256, 56)], 88)


class NetStateC(C.Structure):
    _fields_ = [("n_storage_history", C.c_uint32), ("n_event_history", C.c_uint32), ("n_events", C.c_uint32), ("n_l1_messages", C.c_uint32),
                ("n_final_storage", C.c_uint32), ("reserved0", C.c_uint32), ("storage_history", C.c_void_p), ("event_history", C.c_void_p),
                ("events", C.c_void_p), ("l1_messages", C.c_void_p), ("final_storage", C.c_void_p)]


class InstanceTraceC(C.Structure):
    _fields_ = [("status", C.c_uint32), ("n_cycles", C.c_uint32), ("n_mem", C.c_uint32), ("n_log", C.c_uint32), ("n_aux", C.c_uint32),
                ("reserved0", C.c_uint32), ("records", C.c_void_p), ("mem", C.c_void_p), ("log", C.c_void_p), ("aux", C.c_void_p),
                ("mem_off", C.c_void_p), ("log_off", C.c_void_p), ("aux_off", C.c_void_p), ("final_state", C.c_uint8 * 680)]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _from_ptr(ptr, count, dtype):
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


# ----------------------------------------------------------------------------------------
# ISA helpers (host-only library)
# ----------------------------------------------------------------------------------------
_isa_lib = None


def isa_lib():
    global _isa_lib
    if _isa_lib is None:
        path = _build.ISA_LIB
        if not os.path.exists(path):
            _build.build_isa()
        _isa_lib = C.CDLL(path)
        _isa_lib.zkw_isa_encode.restype = C.c_uint64
        _isa_lib.zkw_isa_encode.argtypes = [C.c_uint32] * 8
        _isa_lib.zkw_isa_find.restype = C.c_int32
        _isa_lib.zkw_isa_find.argtypes = [C.c_void_p] + [C.c_uint32] * 5
    return _isa_lib


class Isa:
    """The ISA table + instruction encoder used by the synthetic tape generator."""

    def __init__(self, table=None):
        if table is None:
            table = np.zeros(1, dtype=ISA_TABLE)
            rc = isa_lib().zkw_isa_default(_ptr(table))
            assert rc == OK
        self.table = table
        self._cache = {}

    @property
    def consts(self):
        return self.table["consts"][0]

    def find(self, opcode, variant=0, src0=MODE_REG, dst0=MODE_REG, flags=0):
        key = (opcode, variant, src0, dst0, flags)
        if key not in self._cache:
            e = self.table["entries"][0]
            m = (e["opcode"] == opcode) & (e["variant"] == variant) & (e["src0_mode"] == src0) & (e["dst0_mode"] == dst0) & (e["flags"] == flags)
            idx = np.flatnonzero(m)
            if len(idx) == 0:
                raise KeyError("no ISA variant for %r" % (key,))
            self._cache[key] = int(idx[0])
        return self._cache[key]

    # truth tables of the eight Conditions over (lt | eq << 1 | gt << 2), in the order of COND_* (cycle.rs:193-209)
    COND_TRUTH = (0xFF, 0xF0, 0xAA, 0xCC, 0xFC, 0xEE, 0x33, 0xFA)

    def cond_field(self, cond):
        """the 3-bit field value that names the logical condition COND_* under this table's condition_lut"""
        lut = int(self.consts["condition_lut"])
        for f in range(8):
            if (lut >> (8 * f)) & 0xFF == self.COND_TRUTH[cond & 7]:
                return f
        raise KeyError("condition %d has no field value in this table" % cond)

    def fwd_code(self, kind):
        """ABI byte of FarCallForwardPageType `kind` (0 UseHeap, 1 ForwardFatPointer, 2 UseAuxHeap) under this table"""
        return (int(self.consts["forwarding_codes"]) >> (8 * kind)) & 0xFF

    @classmethod
    def variant_of_default(cls, seed, permute=True, reprice=False, clip_mode=None, swap_forwarding=False, permute_conditions=False, shift_registers=False):
        """A table that differs from the recalled default in what the absent crates decide: the numbering of
        the 2048 variants, the prices, the clip mode, the forwarding-mode byte codes, the condition a field value names, the
        register conventions of far_call / ret.  Programs built through `enc` / `fwd_code` follow the table."""
        base = cls()
        t = base.table.copy()
        rng = np.random.default_rng(seed)
        c = t["consts"][0]
        if permute:
            perm = rng.permutation(ISA_TABLE_SIZE)
            e = base.table["entries"][0]
            t["entries"][0][perm] = e
            c["nop_variant_idx"] = perm[int(c["nop_variant_idx"])]
            c["panic_variant_idx"] = perm[int(c["panic_variant_idx"])]
        if reprice:
            t["entries"][0]["price"] = t["entries"][0]["price"] * 3 + 5
        if clip_mode is not None:
            c["clip_mode"] = clip_mode
        if swap_forwarding:
            c["forwarding_codes"] = 2 | (7 << 8) | (0 << 16)  # UseHeap = 2, ForwardFatPointer = 7, UseAuxHeap = 0
        if permute_conditions:
            rows = [int(x) for x in rng.permutation(8)]
            lut = int(c["condition_lut"])
            c["condition_lut"] = sum(((lut >> (8 * rows[f])) & 0xFF) << (8 * f) for f in range(8))
        if shift_registers:  # calldata pointer in r2, marker in r1, implicit parameter in r14, reserved range r13 + r15 ...
            c["call_regs"] = 1 | (0 << 8) | (13 << 16)
            c["call_ranges"] = 3 | (11 << 8) | (11 << 16) | (13 << 24)
            c["ret_regs"] = 1 | (0 << 8) | (2 << 16) | (4 << 24)
        out = cls(t)
        always = out.cond_field(COND_ALWAYS)
        c["nop_encoding"] = int(c["nop_variant_idx"]) | (always << 13)
        c["exception_revert_encoding"] = int(c["panic_variant_idx"]) | (always << 13)
        return out

    def canonical_opcode(self, word):
        """an opcode word of this table -> the same instruction under the default table (variant index and condition field
        mapped back); None when the variant index names no instruction"""
        if not hasattr(self, "_canon"):
            d = Isa()
            self._canon = d
        e = self.table["entries"][0][int(word) & 0x7FF]
        try:
            idx = self._canon.find(int(e["opcode"]), int(e["variant"]), int(e["src0_mode"]), int(e["dst0_mode"]), int(e["flags"]))
        except KeyError:
            return None
        lut = int(self.consts["condition_lut"])
        truth = (lut >> (8 * ((int(word) >> 13) & 7))) & 0xFF
        cond = self.COND_TRUTH.index(truth)
        return (int(word) & ~0x7FF & ~(7 << 13)) | idx | (cond << 13)

    def enc(self, opcode, variant=0, src0_mode=MODE_REG, dst0_mode=MODE_REG, flags=0, cond=COND_ALWAYS, src0=0, src1=0, dst0=0, dst1=0, imm0=0, imm1=0):
        idx = self.find(opcode, variant, src0_mode, dst0_mode, flags)
        cond = self.cond_field(cond)
        return (idx & 0x7FF) | ((cond & 7) << 13) | ((src0 & 15) << 16) | ((src1 & 15) << 20) | ((dst0 & 15) << 24) | ((dst1 & 15) << 28) | (
            (imm0 & 0xFFFF) << 32) | ((imm1 & 0xFFFF) << 48)


def pack_code(opcodes):
    """list of u64 encodings -> code words (U256, 4 opcodes each; opcode k of a word is limb 3-k,
    the BE sub-word order of cycle.rs:86-94)."""
    n = (len(opcodes) + 3) // 4
    words = np.zeros((n, 4), dtype="<u8")
    for i, op in enumerate(opcodes):
        words[i // 4, 3 - (i % 4)] = op
    return words


def u256_from_int(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype="<u8")


def u256_to_int(a):
    a = np.asarray(a, dtype="<u8").reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(4))


def address_bytes(v):
    return np.frombuffer(int(v).to_bytes(20, "little"), dtype="u1").copy()


# ----------------------------------------------------------------------------------------
# Backend
# ----------------------------------------------------------------------------------------
class ZkwError(RuntimeError):
    pass


OPT_DEBUG_FLAGS, OPT_RESET_SKIP, OPT_NO_INLINE_DECOMMIT, OPT_DEBUG_SYNC, OPT_NO_GRAPH, OPT_WAVES_PER_GROUP, OPT_LANES_PER_WAVE = 1, 2, 3, 5, 6, 7, 8
OPT_PACK_BLOCKS, OPT_STAGING_BUFFERS, OPT_READ_VALUES, OPT_LINK_FLAGS_OFF, OPT_LINK_SELFCHECK = 9, 10, 11, 12, 13


class Backend:
    def __init__(self, path, prefix):
        if not os.path.exists(path):
            raise ZkwError("shared library %s is missing — run __graft_entry__.build() (the product has no CPU fallback)" % path)
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path
        self.ctx = C.c_void_p()
        f = self.fn("last_error")
        f.restype = C.c_char_p
        f.argtypes = [C.c_void_p]

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def call(self, name, *args):
        rc = self.fn(name)(*args)
        if rc != OK:
            msg = self.fn("last_error")(self.ctx if self.ctx else None)
            raise ZkwError("%s%s -> %d: %s" % (self.prefix, name, rc, (msg or b"").decode()))
        return rc

    def open(self, isa, device=0):
        self.call("ctx_create", C.c_int(device), C.byref(self.ctx))
        self.call("ctx_set_isa", self.ctx, _ptr(isa.table))
        return self

    def close(self):
        if self.ctx:
            self.fn("ctx_destroy")(self.ctx)
            self.ctx = C.c_void_p()

    def set_option(self, option, value):
        """zkw_ctx_set_option: test hooks / ablations of the engine (OPT_*).  The oracle has none of them: a no-op there."""
        if self.prefix != "zkw_":
            return
        self.call("ctx_set_option", self.ctx, C.c_uint32(option), C.c_uint64(value))

    def create_batch(self, workload):
        return Batch(self, workload)

    def blake2s256(self, messages):
        """zkw_blake2s256: BLAKE2s-256 of a list of byte strings (host buffers) -> list of 32-byte digests"""
        n = len(messages)
        data = b"".join(messages)
        offs = np.zeros(n + 1, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum([len(m) for m in messages], dtype=np.uint64)
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
        out = (C.c_uint8 * max(1, 32 * n))()
        self.call("blake2s256", self.ctx, buf, offs.ctypes.data_as(C.c_void_p), C.c_uint32(n), out)
        raw = bytes(out)
        return [raw[32 * i:32 * i + 32] for i in range(n)]

    def reset_many(self, batches, stream=None):
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        self.call("batches_reset", arr, C.c_uint32(len(batches)), C.c_void_p(stream))

    def run_many(self, batches, max_cycles, stream=None):
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        self.call("batches_run", arr, C.c_uint32(len(batches)), C.c_uint32(max_cycles), C.c_void_p(stream))

    def run_many_committing(self, batches, max_cycles, queue_mask, stream=None):
        """zkw_batches_run_committing: the run chains the queue commitments it can (the decommit queue) itself"""
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        self.call("batches_run_committing", arr, C.c_uint32(len(batches)), C.c_uint32(max_cycles), C.c_uint32(queue_mask), C.c_void_p(stream))

    def commit_many(self, batches, queue_mask, stream=None):
        """zkw_batches_commit: the fused commitment launches alone (caller orders the streams)."""
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        self.call("batches_commit", arr, C.c_uint32(len(batches)), C.c_uint32(queue_mask), C.c_void_p(stream))

    def step_many(self, batches, max_cycles, queue_mask=0, stream=None):
        """zkw_batches_step: reset + run + commit of several batches with fused launches (one batch per grid row)."""
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        self.call("batches_step", arr, C.c_uint32(len(batches)), C.c_uint32(max_cycles), C.c_uint32(queue_mask), C.c_void_p(stream))

    def handle_array(self, batches):
        """the batch handles as the C array the fused entry points take (build once, pass as `batches` again: a caller that
        steps the same group over and over does not rebuild it in front of every launch)"""
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        arr.n = len(batches)
        return arr

    def step_prepared_many(self, batches, max_cycles, queue_mask=0, stream=None):
        """zkw_batches_step_prepared: run + commit of batches whose inputs are in place (uploaded / restored earlier)."""
        arr = batches if hasattr(batches, "n") else self.handle_array(batches)
        self.call("batches_step_prepared", arr, C.c_uint32(arr.n), C.c_uint32(max_cycles), C.c_uint32(queue_mask), C.c_void_p(stream))

    def expand_records_many(self, batches, dst_ptrs, instance_stride=0, stream=None, cycle_stride=None):
        """zkw_batches_expand_records: the 512-byte CycleRecords of every instance of the batches, to one device buffer each"""
        if cycle_stride is None:
            cycle_stride = 1 if instance_stride else 0
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        dst = (C.c_void_p * len(batches))(*dst_ptrs)
        self.call("batches_expand_records", arr, C.c_uint32(len(batches)), dst, C.c_uint64(instance_stride), C.c_uint64(cycle_stride), C.c_void_p(stream))


class DeliveredC(C.Structure):
    _fields_ = [("bytes", C.c_uint64), ("pack_ms", C.c_double), ("n_batches", C.c_uint32), ("n_waves", C.c_uint32), ("overflow", C.c_uint32), ("link_flags", C.c_uint32)]


CYCLE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32)


def trace_checksum(t):
    """the fold of zkw_delivery_replay's built-in consumer over one per-instance trace (include/zkw.h): the sum, mod 2^64, over the
    records / memory / log / aux records of sum_j (u64[j] ^ K * (j + 1 + w0)), K = 0x9E3779B97F4A7C15, w0 = 1 / 3 / 5 / 7"""
    acc = 0
    kk = 0x9E3779B97F4A7C15
    for key, words, w0 in (("records", 64, 1), ("mem", 6, 3), ("log", 16, 5), ("aux", 32, 7)):
        a = t[key]
        if len(a) == 0:
            continue
        u = np.frombuffer(a.tobytes(), dtype="<u8").reshape(len(a), words)
        ks = np.array([(kk * (j + 1 + w0)) & 0xFFFFFFFFFFFFFFFF for j in range(words)], dtype=np.uint64)
        with np.errstate(over="ignore"):
            acc += int((u ^ ks).sum(dtype=np.uint64))
    return acc & 0xFFFFFFFFFFFFFFFF


class Delivery:
    """zkw_delivery (include/zkw.h): whole steps delivered into a persistent pinned host ring by one pack kernel per step;
    traces rebuilt / replayed from the ring on a pool of host threads."""

    def __init__(self, backend, n_slots, slot_bytes, host_threads=1):
        self.be = backend
        self.h = C.c_void_p()
        backend.call("delivery_create", backend.ctx, C.c_uint32(n_slots), C.c_uint64(slot_bytes), C.c_uint32(host_threads), C.byref(self.h))
        self._keep = []

    @staticmethod
    def worst_case_bytes(backend, batches):
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        out = C.c_uint64()
        backend.call("delivery_slot_bytes", arr, C.c_uint32(len(batches)), C.byref(out))
        return out.value

    def submit(self, batches, stream=None):
        arr = batches if hasattr(batches, "n") else self.be.handle_array(batches)
        t = C.c_uint32()
        self.be.call("delivery_submit", self.h, arr, C.c_uint32(arr.n), C.c_void_p(stream), C.byref(t))
        return t.value

    def order_after(self, ticket, stream=None):
        self.be.call("delivery_order_after", self.h, C.c_uint32(ticket), C.c_void_p(stream))

    def wait(self, ticket):
        info = DeliveredC()
        self.be.call("delivery_wait", self.h, C.c_uint32(ticket), C.byref(info))
        return {"bytes": info.bytes, "pack_ms": info.pack_ms, "n_batches": info.n_batches, "n_waves": info.n_waves, "overflow": info.overflow, "link_flags": info.link_flags}

    def trace(self, ticket, batch_index, i):
        t = InstanceTraceC()
        self.be.call("delivery_get_instance_trace", self.h, C.c_uint32(ticket), C.c_uint32(batch_index), C.c_uint32(i), C.byref(t))
        return _trace_dict(t)

    def replay(self, ticket, fn=None):
        """-> (cycles, checksum).  fn(thread, batch, instance, cycle, record_ptr, mem_ptr, n_mem, log_ptr, n_log, aux_ptr, n_aux) or
        None for the built-in fold"""
        n, acc = C.c_uint64(), C.c_uint64()
        if fn is None:
            cb = C.cast(None, CYCLE_FN)
        else:
            cb = CYCLE_FN(lambda user, *a: fn(*a))
        self.be.call("delivery_replay", self.h, C.c_uint32(ticket), cb, None, C.byref(n), C.byref(acc))
        return n.value, acc.value

    def release(self, ticket):
        self.be.call("delivery_release", self.h, C.c_uint32(ticket))

    def close(self):
        if self.h:
            self.be.fn("delivery_destroy")(self.h)
            self.h = C.c_void_p()


def _trace_dict(t):
    n = t.n_cycles
    return {
        "status": t.status,
        "n_cycles": n,
        "records": _from_ptr(t.records, n, CYCLE_RECORD),
        "mem": _from_ptr(t.mem, t.n_mem, MEM_QUERY),
        "log": _from_ptr(t.log, t.n_log, LOG_QUERY),
        "aux": _from_ptr(t.aux, t.n_aux, AUX_EVENT),
        "mem_off": _from_ptr(t.mem_off, n + 1, np.dtype("<u4")),
        "log_off": _from_ptr(t.log_off, n + 1, np.dtype("<u4")),
        "aux_off": _from_ptr(t.aux_off, n + 1, np.dtype("<u4")),
        "final_state": np.frombuffer(bytes(t.final_state), dtype=VM_LOCAL_STATE, count=1).copy()[0],
    }


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32)


class Comm:
    """zkw_comm: the communicator of the final exchange (zkw_reduce_commitments, include/zkw.h).

    Comm.rccl(backend, rank, world, id_bytes)   RCCL over xGMI (id from Comm.unique_id(), shared by the caller)
    Comm.external(backend, rank, world, allgather, allreduce_sum)   caller-supplied collectives on host buffers:
        allgather(send: np.uint8[bytes]) -> np.uint8[world * bytes];  allreduce_sum(np.uint64[count]) -> np.uint64[count]
    """

    def __init__(self, backend):
        self.be = backend
        self.h = C.c_void_p()
        self.world = 1
        self._keep = []

    @staticmethod
    def probe(backend):
        """zkw_comm_probe: librccl loads and has the symbols the RCCL transport uses (no side effects); raises ZkwError otherwise"""
        backend.call("comm_probe")
        return True

    @staticmethod
    def unique_id(backend):
        buf = (C.c_uint8 * 128)()
        backend.call("comm_get_unique_id", buf)
        return bytes(buf)

    @classmethod
    def rccl(cls, backend, rank, world, id_bytes):
        self = cls(backend)
        self.world = world
        buf = (C.c_uint8 * 128)(*id_bytes)
        backend.call("comm_create_rccl", backend.ctx, C.c_int(rank), C.c_int(world), buf, C.byref(self.h))
        self.device_buffers = True
        return self

    @classmethod
    def external(cls, backend, rank, world, allgather=None, allreduce_sum=None):
        self = cls(backend)
        self.world = world

        def _ag(user, send, recv, nbytes):
            try:
                s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                out = np.ascontiguousarray(allgather(s.copy()), dtype=np.uint8).reshape(-1)
                assert out.size == nbytes * world
                C.memmove(recv, out.ctypes.data, out.size)
                return 0
            except Exception:  # noqa: BLE001  (reported to C as a failed collective)
                return 1

        def _ar(user, inout, count):
            try:
                a = np.ctypeslib.as_array(inout, shape=(count,))
                a[:] = np.asarray(allreduce_sum(a.copy()), dtype=np.uint64)
                return 0
            except Exception:  # noqa: BLE001
                return 1

        ag = ALLGATHER_FN(_ag) if allgather else C.cast(None, ALLGATHER_FN)
        ar = ALLREDUCE_FN(_ar) if allreduce_sum else C.cast(None, ALLREDUCE_FN)
        self._keep = [ag, ar]
        backend.call("comm_create_external", backend.ctx, C.c_int(rank), C.c_int(world), ag, ar, None, C.byref(self.h))
        self.device_buffers = False
        return self

    def exchange_sizes(self, n_instances, stream=None):
        """zkw_comm_exchange_sizes — COLLECTIVE: every rank announces the shard size it reduces from now on"""
        self.be.call("comm_exchange_sizes", self.h, C.c_uint32(n_instances), C.c_void_p(stream))

    def reduce(self, batches, queue_mask, gathered=None, want_total=False, stream=None):
        """zkw_reduce_commitments.  `gathered`: a device pointer (int) for an RCCL communicator, None or a numpy u64
        array for an external one (allocated here when None).  Returns (gathered, n_max, sizes, total_stats | None)."""
        arr = (C.c_void_p * len(batches))(*[b.h.value for b in batches])
        n_max = C.c_uint32()
        sizes = (C.c_uint32 * self.world)()
        total = np.zeros(1, dtype=RUN_STATS)
        nq = bin(queue_mask & 7).count("1")
        gptr = None
        if self.device_buffers:
            gptr = C.c_void_p(gathered) if gathered else None
        else:
            if gathered is None and nq:
                # rows per rank are only known after the size exchange: ask for it first
                self.be.call("reduce_commitments", self.h, arr, C.c_uint32(len(batches)), C.c_uint32(0), None, C.byref(n_max), sizes, None, C.c_void_p(stream))
                gathered = np.zeros((self.world, len(batches), n_max.value, nq, 4), dtype="<u8")
            gptr = _ptr(gathered) if gathered is not None else None
        self.be.call("reduce_commitments", self.h, arr, C.c_uint32(len(batches)), C.c_uint32(queue_mask), gptr, C.byref(n_max), sizes,
                     _ptr(total) if want_total else None, C.c_void_p(stream))
        return gathered, n_max.value, list(sizes), (total[0] if want_total else None)

    def close(self):
        if self.h:
            self.be.fn("comm_destroy")(self.h)
            self.h = C.c_void_p()


def load_product():
    """libzkw.so — the HIP library. Raises if it has not been built; never falls back."""
    return Backend(_build.LIB, "zkw_")


class Batch:
    """One staged + uploaded batch (see synth.Workload for the inputs)."""

    def __init__(self, backend, wl):
        self.be = backend
        self.wl = wl
        self.h = C.c_void_p()
        limits = np.zeros(1, dtype=LIMITS)
        for k, v in wl.limits.items():
            limits[k] = v
        self.limits = limits
        be = backend
        be.call("batch_create", be.ctx, C.c_uint32(wl.n_instances), _ptr(limits), C.byref(self.h))
        blob_ids = []
        for words in wl.blobs:
            words = np.ascontiguousarray(words, dtype="<u8").reshape(-1, 4)
            bid = C.c_uint32()
            be.call("batch_add_code_blob", self.h, _ptr(words), C.c_uint32(len(words)), C.byref(bid))
            blob_ids.append(bid.value)
        self.blob_ids = blob_ids
        for h, bi in wl.preimages:
            hh = np.ascontiguousarray(h, dtype="<u8")
            be.call("batch_add_decommit_preimage", self.h, _ptr(hh), C.c_uint32(blob_ids[bi]))
        for first, count, page, bi in wl.code_pages:
            be.call("batch_set_code_page", self.h, C.c_uint32(first), C.c_uint32(count), C.c_uint32(page), C.c_uint32(blob_ids[bi]))
        states = np.ascontiguousarray(wl.states)
        inner = np.ascontiguousarray(wl.inner)
        assert states.dtype == VM_LOCAL_STATE and inner.dtype == CALLSTACK_ENTRY
        depth = inner.shape[1] if inner.ndim == 2 else 0
        be.call("batch_set_state", self.h, C.c_uint32(0), C.c_uint32(wl.n_instances), _ptr(states), _ptr(inner), C.c_uint32(depth))
        if wl.heaps is not None:
            heaps = np.ascontiguousarray(wl.heaps, dtype="<u8")  # [n, words, 4]
            lens = getattr(wl, "heap_lens", None)  # (optional: a shorter image per instance — words beyond it read as zero)
            for i in range(wl.n_instances):
                be.call("batch_set_heap", self.h, C.c_uint32(i), _ptr(heaps[i]), C.c_uint32(heaps.shape[1] if lens is None else int(lens[i])))
        if getattr(wl, "bootloader_calldata", None) is not None:
            cd = np.ascontiguousarray(wl.bootloader_calldata, dtype="<u8")  # [n, words, 4]
            for i in range(wl.n_instances):
                be.call("batch_set_bootloader_calldata", self.h, C.c_uint32(i), _ptr(cd[i]), C.c_uint32(cd.shape[1]))
        if wl.storage is not None:
            for i in range(wl.n_instances):
                s = np.ascontiguousarray(wl.storage[i])
                if len(s):
                    assert s.dtype == STORAGE_SLOT
                    be.call("batch_set_storage", self.h, C.c_uint32(i), _ptr(s), C.c_uint32(len(s)))
        props = np.zeros(1, dtype=BLOCK_PROPERTIES)
        props["default_aa_code_hash"][0] = wl.default_aa_code_hash
        props["zkporter_is_available"] = wl.zkporter_is_available
        be.call("batch_set_block_properties", self.h, _ptr(props))
        be.call("batch_upload", self.h)

    def reset(self, stream=None):
        self.be.call("batch_reset", self.h, C.c_void_p(stream))

    def run(self, max_cycles, stream=None):
        self.be.call("batch_run", self.h, C.c_uint32(max_cycles), C.c_void_p(stream))

    def sync(self):
        self.be.call("batch_sync", self.h)

    def stats(self):
        st = np.zeros(1, dtype=RUN_STATS)
        self.be.call("batch_get_stats", self.h, _ptr(st))
        return st[0]

    def trace(self, i):
        t = InstanceTraceC()
        self.be.call("batch_get_instance_trace", self.h, C.c_uint32(i), C.byref(t))
        return _trace_dict(t)

    def staging(self):
        """zkw_batch_staging: numpy views of the batch's pinned staging buffers — states [n] and heap images [n, words, 4] — for a
        caller that builds its next inputs in place (restage(*batch.staging()) then copies nothing on the host)"""
        sp, hp, nh = C.c_void_p(), C.c_void_p(), C.c_uint32()
        self.be.call("batch_staging", self.h, C.byref(sp), C.byref(hp), C.byref(nh))
        n = self.wl.n_instances
        states = np.frombuffer((C.c_uint8 * (n * VM_LOCAL_STATE.itemsize)).from_address(sp.value), dtype=VM_LOCAL_STATE, count=n)
        heaps = None
        if nh.value:
            heaps = np.frombuffer((C.c_uint8 * (n * nh.value * 32)).from_address(hp.value), dtype="<u8").reshape(n, nh.value, 4)
        return states, heaps

    def restage(self, states, heaps=None, stream=None):
        """zkw_batch_restage: new VmLocalStates (and heap images) for every instance of the uploaded batch, asynchronously on
        `stream`; the batch is restored to them (as after zkw_batch_reset)"""
        states = np.ascontiguousarray(states)
        assert states.dtype == VM_LOCAL_STATE and len(states) == self.wl.n_instances
        hw, nh = None, 0
        if heaps is not None:
            heaps = np.ascontiguousarray(heaps, dtype="<u8")  # [n, words, 4]
            hw, nh = _ptr(heaps), heaps.shape[1]
        self.be.call("batch_restage", self.h, _ptr(states), hw, C.c_uint32(nh), C.c_void_p(stream))

    def page(self, i, page, first_word, n_words):
        """`vm.memory.dump_page_content_as_u256_words(page, first..first + n)` of instance i after the run
        (reference_impls/memory.rs:316-396): [n_words, 4] little-endian u64 limbs."""
        out = np.zeros((max(n_words, 1), 4), dtype="<u8")
        self.be.call("batch_get_page", self.h, C.c_uint32(i), C.c_uint32(page), C.c_uint32(first_word), C.c_uint32(n_words), _ptr(out))
        return out[:n_words]

    def net_state(self, i):
        """get_final_net_states (testing/mod.rs:42-71) of instance i: storage / event histories, net events and L1
        messages, final storage."""
        t = NetStateC()
        self.be.call("batch_get_net_state", self.h, C.c_uint32(i), C.byref(t))
        return {
            "storage_history": _from_ptr(t.storage_history, t.n_storage_history, LOG_QUERY),
            "event_history": _from_ptr(t.event_history, t.n_event_history, LOG_QUERY),
            "events": _from_ptr(t.events, t.n_events, EVENT_MESSAGE),
            "l1_messages": _from_ptr(t.l1_messages, t.n_l1_messages, EVENT_MESSAGE),
            "final_storage": _from_ptr(t.final_storage, t.n_final_storage, STORAGE_SLOT),
        }

    def expand_records(self, first, count, dst_device, instance_stride=0, stream=None, cycle_stride=None):
        """zkw_batch_expand_records: the 512-byte CycleRecords of instances [first, first + count) written to device memory;
        record (i, k) at ((i - first) * instance_stride + k * cycle_stride) * 512 (default: instance-major)"""
        if cycle_stride is None:
            cycle_stride = 1 if instance_stride else 0
        self.be.call("batch_expand_records", self.h, C.c_uint32(first), C.c_uint32(count), C.c_void_p(dst_device), C.c_uint64(instance_stride), C.c_uint64(cycle_stride), C.c_void_p(stream))

    def commitments(self):
        out = np.zeros((self.wl.n_instances, QUEUE_COUNT, 4), dtype="<u8")
        self.be.call("batch_get_commitments", self.h, _ptr(out))
        return out

    def destroy(self):
        if self.h:
            self.be.fn("batch_destroy")(self.h)
            self.h = C.c_void_p()


TRACE_ARRAYS = ("records", "mem", "log", "aux", "mem_off", "log_off", "aux_off")


def net_states_equal(a, b):
    for k in ("storage_history", "event_history", "events", "l1_messages", "final_storage"):
        if a[k].shape != b[k].shape:
            return False, "%s: shape %r != %r" % (k, a[k].shape, b[k].shape)
        if a[k].tobytes() != b[k].tobytes():
            for j in range(len(a[k])):
                if a[k][j].tobytes() != b[k][j].tobytes():
                    return False, "%s[%d]: %r vs %r" % (k, j, a[k][j], b[k][j])
    return True, ""


def traces_equal(a, b):
    """Bit-exact comparison of two per-instance traces; returns (ok, first difference)."""
    for k in ("status", "n_cycles"):
        if a[k] != b[k]:
            return False, "%s: %r != %r" % (k, a[k], b[k])
    for k in TRACE_ARRAYS:
        x, y = a[k], b[k]
        if x.shape != y.shape:
            return False, "%s: shape %r != %r" % (k, x.shape, y.shape)
        if x.tobytes() != y.tobytes():
            xb = np.frombuffer(x.tobytes(), dtype="u1").reshape(len(x), -1)
            yb = np.frombuffer(y.tobytes(), dtype="u1").reshape(len(y), -1)
            row = int(np.flatnonzero((xb != yb).any(axis=1))[0])
            col = int(np.flatnonzero(xb[row] != yb[row])[0])
            return False, "%s[%d] differs at byte %d: %r vs %r" % (k, row, col, x[row], y[row])
    if a["final_state"].tobytes() != b["final_state"].tobytes():
        return False, "final_state: %r vs %r" % (a["final_state"], b["final_state"])
    return True, ""
