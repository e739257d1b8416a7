/*
 * zkw.h — C ABI of the MI355X-native out-of-circuit EraVM witness generator.
 *
 * This header is the drop-in boundary.  The reference crate
 * (matter-labs/era-zk_evm, `zk_evm` v1.4.1) has no FFI of its own: its boundary is the
 * Rust generic `VmState<S,M,EV,PP,DP,WT,N,E>` (reference src/vm_state/mod.rs:157-175)
 * whose only entry point is `cycle()` (src/vm_state/cycle.rs:257-429).  A Rust shim that
 * keeps that trait surface binds the functions below (INTEGRATION.md shows the stub);
 * the C++ mirror in era-zk_evm_amd/host/ and the Python ctypes loader bind the same
 * symbols.  Every struct here is plain-old-data, little-endian, naturally aligned.
 *
 * Data-model summary
 *   U256            4 x u64 little-endian limbs (`.0[0]` lowest; mul.rs:36-39, uma.rs:337)
 *   Address         20 bytes, little-endian integer (byte 0 = least significant); the
 *                   Rust shim converts to/from H160 big-endian bytes
 *   one batch     = N independent VM instances (each what the reference calls a VmState
 *                   with its own Memory/Storage/Decommitter oracles, mod.rs:167-174)
 *   one run       = every instance executes `cycle()` until execution_has_ended()
 *                   (mod.rs:214-216) or max_cycles
 *   output        = per instance: one 512-byte CycleRecord per executed cycle + three
 *                   ordered query logs (memory / log / aux), which together carry exactly
 *                   what the 10 VmWitnessTracer callbacks receive
 *                   (src/witness_trace/mod.rs:11-72), in SURVEY Appendix-A order.
 */
#ifndef ZKW_H
#define ZKW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------ */
/* Scalars                                                                               */
/* ------------------------------------------------------------------------------------ */

typedef struct zkw_u256 {
  uint64_t l[4];
} zkw_u256;

#define ZKW_REGISTERS_COUNT 15 /* zkevm_opcode_defs::REGISTERS_COUNT; r0 is constant zero (helpers.rs:318-334) */

/* return codes of every entry point */
#define ZKW_OK 0
#define ZKW_ERR_INVALID (-1)  /* bad argument / call order                           */
#define ZKW_ERR_DEVICE (-2)   /* HIP runtime error (no GPU, OOM, launch failure)      */
#define ZKW_ERR_LIMIT (-3)    /* a zkw_limits capacity was exceeded while staging     */
#define ZKW_ERR_NOT_RUN (-4)  /* results requested before a run                       */

/* per-instance status after a run (reference panics / Err map here) */
#define ZKW_STATUS_RUNNING 0           /* stopped at max_cycles, execution not ended          */
#define ZKW_STATUS_ENDED 1             /* callstack empty: execution_has_ended()              */
#define ZKW_STATUS_UNKNOWN_CODE_HASH 2 /* decommitter.rs:54-56 -> cycle() returned Err       */
#define ZKW_STATUS_REFERENCE_PANIC 3   /* an assert!/unwrap/expect of the reference would fire */
#define ZKW_STATUS_LIMIT 4             /* a zkw_limits capacity was exceeded on device        */

/* ------------------------------------------------------------------------------------ */
/* ISA table (uploaded by the host)                        */
/* ------------------------------------------------------------------------------------ */

/* opcode families, order of zkevm_opcode_defs::Opcode / parsing.rs:61-78 */
enum {
  ZKW_OP_INVALID = 0,
  ZKW_OP_NOP = 1,
  ZKW_OP_ADD = 2,
  ZKW_OP_SUB = 3,
  ZKW_OP_MUL = 4,
  ZKW_OP_DIV = 5,
  ZKW_OP_JUMP = 6,
  ZKW_OP_CONTEXT = 7,
  ZKW_OP_SHIFT = 8,
  ZKW_OP_BINOP = 9,
  ZKW_OP_PTR = 10,
  ZKW_OP_NEAR_CALL = 11,
  ZKW_OP_LOG = 12,
  ZKW_OP_FAR_CALL = 13,
  ZKW_OP_RET = 14,
  ZKW_OP_UMA = 15
};

/* inner variants */
enum { ZKW_CTX_THIS = 0, ZKW_CTX_CALLER, ZKW_CTX_CODE_ADDRESS, ZKW_CTX_META, ZKW_CTX_ERGS_LEFT, ZKW_CTX_SP,
       ZKW_CTX_GET_CONTEXT_U128, ZKW_CTX_SET_CONTEXT_U128, ZKW_CTX_SET_ERGS_PER_PUBDATA, ZKW_CTX_INC_TX_NUMBER };
enum { ZKW_SHIFT_SHL = 0, ZKW_SHIFT_SHR, ZKW_SHIFT_ROL, ZKW_SHIFT_ROR };
enum { ZKW_BINOP_XOR = 0, ZKW_BINOP_AND, ZKW_BINOP_OR };
enum { ZKW_PTR_ADD = 0, ZKW_PTR_SUB, ZKW_PTR_PACK, ZKW_PTR_SHRINK };
enum { ZKW_LOG_STORAGE_READ = 0, ZKW_LOG_STORAGE_WRITE, ZKW_LOG_TO_L1, ZKW_LOG_EVENT, ZKW_LOG_PRECOMPILE };
enum { ZKW_FAR_NORMAL = 0, ZKW_FAR_DELEGATE, ZKW_FAR_MIMIC };
enum { ZKW_RET_OK = 0, ZKW_RET_REVERT, ZKW_RET_PANIC };
enum { ZKW_UMA_HEAP_READ = 0, ZKW_UMA_HEAP_WRITE, ZKW_UMA_AUX_READ, ZKW_UMA_AUX_WRITE, ZKW_UMA_FAT_PTR_READ };

/* operand addressing modes = ImmMemHandlerFlags (mem_ops.rs:37-122) */
enum {
  ZKW_MODE_REG = 0,        /* RegOnly | Full(UseRegOnly) | RegOrImm(UseRegOnly) */
  ZKW_MODE_STACK_PP = 1,   /* UseStackWithPushPop   */
  ZKW_MODE_STACK_OFF = 2,  /* UseStackWithOffset    */
  ZKW_MODE_STACK_ABS = 3,  /* UseAbsoluteOnStack    */
  ZKW_MODE_IMM = 4,        /* UseImm16Only (src only) */
  ZKW_MODE_CODE = 5        /* UseCodePage  (src only) */
};

/* zkw_isa_entry.props bits = the per-variant accessors cycle.rs uses */
#define ZKW_PROP_EXPLICIT_PANIC 0x01  /* variant.is_explicit_panic()            cycle.rs:142 */
#define ZKW_PROP_KERNEL_ONLY 0x02     /* variant.requires_kernel_mode()         cycle.rs:174 */
#define ZKW_PROP_STATIC_OK 0x04       /* variant.can_be_used_in_static_context() cycle.rs:178 */
#define ZKW_PROP_SWAP 0x08            /* variant.swap_operands()                cycle.rs:341 */
#define ZKW_PROP_SRC0_PTR_OK 0x10     /* opcode.src0_can_be_pointer()           cycle.rs:375 */
#define ZKW_PROP_SRC1_PTR_OK 0x20     /* opcode.src1_can_be_pointer()           cycle.rs:386 */

typedef struct zkw_isa_entry {
  uint8_t opcode;    /* ZKW_OP_*                         */
  uint8_t variant;   /* inner variant of the family       */
  uint8_t src0_mode; /* ZKW_MODE_*                        */
  uint8_t dst0_mode; /* ZKW_MODE_REG..ZKW_MODE_STACK_ABS  */
  uint8_t flags;     /* bit i = variant.flags[i]          */
  uint8_t props;     /* ZKW_PROP_*                        */
  uint16_t reserved;
  uint32_t price;    /* OPCODES_PRICES[idx]  cycle.rs:147-148 */
} zkw_isa_entry;

#define ZKW_ISA_TABLE_SIZE 2048 /* 1 << OPCODES_TABLE_WIDTH (11) */

/* scalar constants of zkevm_opcode_defs / zk_evm_abstractions the path reads (Appendix B) */
typedef struct zkw_isa_consts {
  uint64_t nop_encoding;              /* E::nop_encoding()               cycle.rs:126 */
  uint64_t exception_revert_encoding; /* E::exception_revert_encoding()  cycle.rs:115 */
  uint32_t panic_variant_idx;         /* variant DecodedOpcode::mask_into_panic() installs */
  uint32_t nop_variant_idx;           /* variant DecodedOpcode::mask_into_nop() installs   */
  uint32_t clip_mode;                 /* AllowedPcOrImm::from_u64_clipped: 0 = saturate at 0xffff, 1 = truncate */
  uint32_t time_delta_per_cycle;      /* TIME_DELTA_PER_CYCLE (4)                mod.rs:233 */
  uint32_t new_memory_pages_per_far_call; /* NEW_MEMORY_PAGES_PER_FAR_CALL (8)   mod.rs:239 */
  uint32_t vm_max_stack_depth;        /* VM_MAX_STACK_DEPTH          execution_stack.rs:120 */
  uint32_t initial_sp_on_far_call;    /* INITIAL_SP_ON_FAR_CALL (0)      far_call.rs:543    */
  uint32_t new_frame_memory_stipend;  /* NEW_FRAME_MEMORY_STIPEND (4096) far_call.rs:553    */
  uint32_t memory_growth_ergs_per_byte; /* MEMORY_GROWTH_ERGS_PER_BYTE (1) uma.rs:197       */
  uint32_t ergs_per_code_word_decommittment; /* ERGS_PER_CODE_WORD_DECOMMITTMENT far_call.rs:424 */
  uint32_t initial_storage_write_pubdata_bytes; /* log.rs:107 */
  uint32_t l1_message_pubdata_bytes;  /* log.rs:123 */
  uint32_t max_offset_to_deref_low;   /* uma::MAX_OFFSET_TO_DEREF = 2^32-33      uma.rs:127 */
  uint32_t deployer_address_low;      /* DEPLOYER_SYSTEM_CONTRACT_ADDRESS (0x8002) far_call.rs:136 */
  uint32_t keccak_precompile_address; /* 0x8010 */
  uint32_t sha256_precompile_address; /* 0x02   */
  uint32_t ecrecover_precompile_address; /* 0x01 */
  uint8_t storage_aux_byte, event_aux_byte, l1_message_aux_byte, precompile_aux_byte; /* log.rs:6-8 */
  uint32_t ecrecover_input_layout;    /* words at input_memory_offset: 0 = (hash, r, s, v) as the reference's own test fills
                                         memory (testing/tests/precompiles/ecrecover.rs:3-49); 1 = (hash, v, r, s) */
  uint32_t bootloader_calldata_page;  /* zkevm_opcode_defs::BOOTLOADER_CALLDATA_PAGE (memory.rs:11,229-231): the page that
                                         SimpleMemory keeps in `pages_with_extended_lifetime` from the start.  Value recalled,
                                         UNVERIFIED (the crate is not on disk) — hence a table constant, not code: a caller that
                                         relies on zkw_batch_get_page for this page supplies the real value (the shim's
                                         isa_from_opcode_defs does); a wrong value only moves which page number the words of
                                         zkw_batch_set_bootloader_calldata are reported under, the VM cannot read the page */
  /* Conventions of the two absent crates that rounds 1-3 had compiled in (SURVEY App. B lists them as recalled): table
   * constants since round 4, so that a shim built against the real zkevm_opcode_defs supplies the real values. */
  uint32_t call_regs;                 /* far_call.rs:506-508,573-610, INDICES into VmLocalState.registers (r1 = 0): byte 0
                                         CALL_IMPLICIT_CALLDATA_FAT_PTR_REGISTER (0), byte 1 CALL_IMPLICIT_CONSTRUCTOR_MARKER_REGISTER (1),
                                         byte 2 CALL_IMPLICIT_PARAMETER_REG_IDX (14) */
  uint32_t call_ranges;               /* byte 0 / 1: CALL_SYSTEM_ABI_REGISTERS first / end (2 / 12), byte 2 / 3: CALL_RESERVED_RANGE
                                         first / end (12 / 14); ends exclusive */
  uint32_t ret_regs;                  /* ret.rs:213-233: byte 0 RET_IMPLICIT_RETURNDATA_PARAMS_REGISTER (0), bytes 1..3
                                         RET_RESERVED_REGISTER_0..2 (1, 2, 3); every register behind the last one is cleared */
  uint32_t forwarding_codes;          /* FarCallForwardPageType as the ABI byte (far_call.rs:255, ret.rs:59): byte 0 UseHeap (0),
                                         byte 1 ForwardFatPointer (1), byte 2 UseAuxHeap (2); any other byte decodes as UseHeap */
  uint32_t unmapped_page;             /* UNMAPPED_PAGE (0)  far_call.rs:162,439 */
  uint32_t reserved0;
  uint64_t max_offset_for_add_sub;    /* ptr::MAX_OFFSET_FOR_ADD_SUB (2^32)  ptr.rs:47,111: an offset >= this panics */
  uint64_t condition_lut;             /* Condition, as the 3-bit field of the opcode word decodes (cycle.rs:193-209): bit
                                         8 * field + (lt_of | eq << 1 | gt << 2) = the condition holds.  Default order Always,
                                         Gt, Lt, Eq, Ge, Le, Ne, GtOrLt = 0xfa33eefcccaaf0ff */
} zkw_isa_consts;

typedef struct zkw_isa_table {
  zkw_isa_entry entries[ZKW_ISA_TABLE_SIZE];
  zkw_isa_consts consts;
} zkw_isa_table;

/* ------------------------------------------------------------------------------------ */
/* VM state (input: initial state; output: final state)                                  */
/* ------------------------------------------------------------------------------------ */

/* CallStackEntry, execution_stack.rs:6-24 */
typedef struct zkw_callstack_entry {
  uint8_t this_address[20];
  uint8_t msg_sender[20];
  uint8_t code_address[20];
  uint32_t base_memory_page;
  uint32_t code_page;
  uint16_t sp;
  uint16_t pc;
  uint16_t exception_handler_location;
  uint8_t is_static;
  uint8_t is_local_frame;
  uint32_t ergs_remaining;
  uint8_t this_shard_id;
  uint8_t caller_shard_id;
  uint8_t code_shard_id;
  uint8_t reserved0;
  uint32_t reserved1;
  uint64_t context_u128_value[2]; /* little-endian u128 */
  uint32_t heap_bound;
  uint32_t aux_heap_bound;
} zkw_callstack_entry; /* 112 bytes */

/* VmLocalState, mod.rs:54-73 (callstack.inner passed separately) */
typedef struct zkw_vm_local_state {
  zkw_u256 previous_code_word;
  zkw_u256 registers[ZKW_REGISTERS_COUNT];
  uint16_t register_ptr_bitmap; /* bit i = registers[i].is_pointer */
  uint8_t flags;                /* bit0 overflow_or_less_than, bit1 equality, bit2 greater_than (flags.rs:4-8) */
  uint8_t pending_exception;
  uint32_t previous_code_memory_page;
  uint32_t timestamp;
  uint32_t monotonic_cycle_counter;
  uint32_t spent_pubdata_counter;
  uint32_t memory_page_counter;
  uint32_t absolute_execution_step;
  uint32_t current_ergs_per_pubdata_byte;
  uint16_t tx_number_in_block;
  uint16_t previous_super_pc;
  uint32_t callstack_depth; /* callstack.inner.len(); 0 => execution_has_ended() */
  uint64_t context_u128_register[2];
  zkw_callstack_entry current; /* callstack.current */
} zkw_vm_local_state; /* 512 + 4 + 28 + 4 + 4 + 16 + 112 = 680 bytes */

/* BlockProperties, block_properties/mod.rs:4-7 */
typedef struct zkw_block_properties {
  zkw_u256 default_aa_code_hash;
  uint32_t zkporter_is_available;
  uint32_t reserved0;
} zkw_block_properties;

/* one pre-populated storage slot, testing/storage.rs:25-31 `populate` */
typedef struct zkw_storage_slot {
  zkw_u256 key;
  zkw_u256 value;
  uint8_t address[20];
  uint8_t shard_id;
  uint8_t reserved0[3];
} zkw_storage_slot; /* 88 bytes */

/* capacities of one batch; every per-instance arena is sized from these.  An instance that WRITES beyond a capacity
 * (or emits more records than a stream holds) stops with ZKW_STATUS_LIMIT; reading a stack / heap word that was never
 * written is not an overrun — it reads zero, as in the reference's zero-filled pages (memory.rs:427-473). */
typedef struct zkw_limits {
  uint32_t max_cycles;            /* cycles recorded per instance and per run                      */
  uint32_t max_far_frames;        /* far-call frames (incl. the bootloader frame) that are live or reachable at one time: a
                                     frame's arena slot is reused once it has returned and its heap / aux heap is no
                                     returndata any more (the reference's page pools, memory.rs:660-758).  It does NOT cap
                                     the code hashes an instance decommits: SimpleDecommitter's history (decommitter.rs:38-47)
                                     has one row per registered (hash -> blob) pair, so a run may decommit every known hash */
  uint32_t max_callstack_depth;   /* near + far frames alive at once                                */
  uint32_t stack_words;           /* words per stack page   (reference: 65536, memory.rs:177-179)  */
  uint32_t heap_words;            /* words per heap page    (reference: grows on demand)            */
  uint32_t aux_heap_words;        /* words per aux-heap page                                        */
  uint32_t storage_slots;         /* distinct (shard,address,key) per instance, power of two        */
  uint32_t storage_journal;       /* storage writes per instance and per run                        */
  uint32_t max_mem_queries;       /* MemoryQuery records per instance and per run; 0 = 6*max_cycles */
  uint32_t max_log_queries;       /* LogQuery records per instance and per run; 0 = max_cycles/2+16 */
  uint32_t max_aux_events;        /* aux events per instance and per run; 0 = derived               */
  uint32_t lanes_per_wave;        /* 0 = the library chooses at upload: full waves (64) for instances that share their code, thin waves for instances that were given different code; 1..64 fixes it.
                                     Thin waves (<= 8) are the geometry of a caller after latency, not throughput: more waves for the same
                                     instances, and every wave gets helper waves that run its keccak256 calls across lanes (a lone batch of
                                     512 precompile-heavy instances: 9.9 ms with full waves, 3.4 ms with lanes_per_wave = 2) */
  uint32_t max_reg_deltas;        /* register writes recorded per instance and per run; 0 = 2*max_cycles + 32 */
  uint32_t reserved[3];
} zkw_limits;

/* ------------------------------------------------------------------------------------ */
/* Witness trace records (output)                                                        */
/* ------------------------------------------------------------------------------------ */

/* CycleRecord: the state after `end_execution_cycle` of one cycle (cycle.rs:413); the state
 * before cycle k is the record of cycle k-1 (or the initial state).  512 bytes:
 *   [0,480)   registers[15] values
 *   [480,512) zkw_cycle_tail
 * Everything else of VmLocalState is either constant inside a frame (restored from the
 * FRAME_START/FRAME_FINISH aux events), derivable (monotonic_cycle_counter = initial + k + 1,
 * previous_code_word = value of the cycle's code read, previous_code_memory_page = code page
 * current at cycle start) or rare (COLD_STATE aux event).
 * On the device a record is stored losslessly as its 32-byte tail plus the values of the registers the cycle wrote
 * (about 70 bytes per cycle instead of 512); zkw_batch_get_instance_trace materialises the full records. */
typedef struct zkw_cycle_tail {
  uint16_t register_ptr_bitmap;
  uint8_t flags;  /* bits 0..2 = lt,eq,gt; bit 3 = pending_exception */
  uint8_t reserved0;
  uint16_t pc;    /* current frame after the cycle */
  uint16_t sp;
  uint32_t ergs_remaining;
  uint32_t timestamp;
  uint32_t heap_bound;
  uint32_t aux_heap_bound;
  uint16_t callstack_depth;
  uint16_t previous_super_pc;
  uint32_t event_counts; /* bits 0-7 mem queries, 8-15 log records, 16-23 aux events emitted this cycle (saturating) */
} zkw_cycle_tail; /* 32 bytes */

typedef struct zkw_cycle_record {
  zkw_u256 registers[ZKW_REGISTERS_COUNT];
  zkw_cycle_tail tail;
} zkw_cycle_record; /* 512 bytes */

/* MemoryType (zk_evm_abstractions::vm::MemoryType) */
enum { ZKW_MEM_STACK = 0, ZKW_MEM_CODE = 1, ZKW_MEM_HEAP = 2, ZKW_MEM_AUX_HEAP = 3, ZKW_MEM_FAT_PTR = 4 };

/* zkw_mem_query.meta bits */
#define ZKW_MQ_TYPE_MASK 0x07 /* ZKW_MEM_*                                             */
#define ZKW_MQ_IS_PTR 0x08    /* value_is_pointer                                      */
#define ZKW_MQ_RW 0x10        /* rw_flag (1 = write)                                   */
#define ZKW_MQ_KIND_SHIFT 5   /* 0 = add_memory_query; 1 = precompile read; 2 = precompile write */

/* MemoryQuery {timestamp, location{memory_type,page,index}, value, value_is_pointer, rw_flag}
 * (field set pinned by helpers.rs:26-32) */
typedef struct zkw_mem_query {
  uint32_t timestamp;
  uint32_t page;
  uint32_t index;
  uint8_t lane; /* device stream only: lane of the owning instance inside its wave; 0 in per-instance views */
  uint8_t seq;  /* order of this record among all records of its instance in its cycle (saturates at 255) */
  uint8_t meta;
  uint8_t reserved0;
  zkw_u256 value;
} zkw_mem_query; /* 48 bytes */

/* zkw_log_query.kind */
enum {
  ZKW_LQ_LOG = 0,    /* WT.add_log_query                                      */
  ZKW_LQ_REFUND = 1  /* WT.record_refund_for_query (log.rs:99-102); refund in `refund_*` */
};
#define ZKW_LQ_RW 0x01
#define ZKW_LQ_ROLLBACK 0x02
#define ZKW_LQ_IS_SERVICE 0x04

/* LogQuery (field set pinned by log.rs:85-97) */
typedef struct zkw_log_query {
  zkw_u256 key;
  zkw_u256 read_value;
  zkw_u256 written_value;
  uint8_t address[20];
  uint32_t timestamp;
  uint16_t tx_number_in_block;
  uint8_t aux_byte;
  uint8_t shard_id;
  uint8_t bools; /* ZKW_LQ_RW | ZKW_LQ_ROLLBACK | ZKW_LQ_IS_SERVICE */
  uint8_t kind;  /* ZKW_LQ_* */
  uint8_t lane;
  uint8_t seq;
} zkw_log_query; /* 128 bytes */

/* aux events */
enum {
  ZKW_AUX_FRAME_START = 1,  /* WT.start_new_execution_context(cc, &prev, &new)  helpers.rs:237-241 */
  ZKW_AUX_FRAME_FINISH = 2, /* WT.finish_execution_context(cc, panicked)        helpers.rs:258-259 */
  ZKW_AUX_DECOMMIT = 3,     /* DP.decommit_into_memory result (+ WT.add_decommittment) helpers.rs:164-194 */
  ZKW_AUX_COLD_STATE = 4    /* rare VmLocalState fields after this cycle                            */
};

typedef struct zkw_aux_event {
  uint8_t type; /* ZKW_AUX_* */
  uint8_t lane;
  uint8_t seq;
  uint8_t flag; /* FRAME_START: 1 = far call; FRAME_FINISH: panicked; DECOMMIT: is_fresh */
  uint32_t a;   /* DECOMMIT: timestamp;   COLD_STATE: spent_pubdata_counter          */
  uint32_t b;   /* DECOMMIT: memory_page; COLD_STATE: current_ergs_per_pubdata_byte  */
  uint32_t c;   /* DECOMMIT: decommitted_length | code blob id << 16; COLD_STATE: tx_number_in_block */
  union {
    struct {
      zkw_callstack_entry previous; /* callstack.current at the call, i.e. what gets pushed */
      zkw_callstack_entry next;
    } frame;                         /* FRAME_START  */
    zkw_u256 hash;                   /* DECOMMIT     */
    struct {
      zkw_u256 hash;
      uint32_t preimage_index;       /* index of the (hash -> blob) pair in the order of zkw_batch_add_decommit_preimage */
    } decommit;                      /* DECOMMIT, with the bookkeeping field the commitment kernels use */
    struct {
      uint64_t context_u128_register[2];
      uint32_t memory_page_counter;
    } cold;                          /* COLD_STATE   */
    uint8_t raw[240];
  } u;
} zkw_aux_event; /* 256 bytes */

/* per-instance view of a finished run; arrays are library-owned, valid until the next
 * run/reset/destroy of the batch.  `*_off[k]..*_off[k+1]` are the records of cycle k. */
typedef struct zkw_instance_trace {
  uint32_t status;   /* ZKW_STATUS_* */
  uint32_t n_cycles; /* executed cycles */
  uint32_t n_mem, n_log, n_aux;
  uint32_t reserved0;
  const zkw_cycle_record* records; /* [n_cycles] */
  const zkw_mem_query* mem;        /* [n_mem] */
  const zkw_log_query* log;        /* [n_log] */
  const zkw_aux_event* aux;        /* [n_aux] */
  const uint32_t* mem_off;         /* [n_cycles + 1] */
  const uint32_t* log_off;         /* [n_cycles + 1] */
  const uint32_t* aux_off;         /* [n_cycles + 1] */
  zkw_vm_local_state final_state;  /* VmLocalState after the last executed cycle */
} zkw_instance_trace;

/* aggregate counters of a run (also the payload of the multi-GPU all-reduce) */
typedef struct zkw_run_stats {
  uint64_t cycles;      /* sum over instances of executed cycles */
  uint64_t mem_queries;
  uint64_t log_queries;
  uint64_t aux_events;
  uint64_t instances_ended;
  uint64_t instances_failed; /* status >= ZKW_STATUS_UNKNOWN_CODE_HASH */
  double kernel_ms;     /* device time of the last run's cycle kernel (HIP events on the run stream) */
  uint64_t reg_deltas;  /* register values written to the delta stream (CycleRecords are stored as tail + deltas) */
} zkw_run_stats;

/* ------------------------------------------------------------------------------------ */
/* Entry points                                                                          */
/* ------------------------------------------------------------------------------------ */

typedef struct zkw_ctx zkw_ctx;     /* one per (thread, GPU); not thread-safe */
typedef struct zkw_batch zkw_batch; /* N instances + their arenas + output streams */

/* Fills `out` with the build's recollection of zkevm_opcode_defs v1.4.1 (UNVERIFIED, the
 * crate is not on disk — SURVEY Appendix B).  A Rust shim overrides it with the real
 * OPCODES_TABLE / OPCODES_PRICES.  Host-only, needs no GPU. */
int zkw_isa_default(zkw_isa_table* out);
/* Encodes one instruction in EncodingModeProduction layout (bits 0-10 variant, 13-15
 * condition, 16-31 src0|src1|dst0|dst1, 32-47 imm0, 48-63 imm1). Host-only. */
uint64_t zkw_isa_encode(uint32_t variant_idx, uint32_t condition, uint32_t src0, uint32_t src1, uint32_t dst0,
                        uint32_t dst1, uint32_t imm0, uint32_t imm1);
/* Finds the variant index of (opcode, variant, src0_mode, dst0_mode, flags); -1 if absent. Host-only. */
int32_t zkw_isa_find(const zkw_isa_table* t, uint32_t opcode, uint32_t variant, uint32_t src0_mode, uint32_t dst0_mode,
                     uint32_t flags);

int zkw_ctx_create(int device, zkw_ctx** out);
void zkw_ctx_destroy(zkw_ctx* ctx);
const char* zkw_last_error(zkw_ctx* ctx); /* ctx may be NULL: last error of a failed zkw_ctx_create */
int zkw_ctx_set_isa(zkw_ctx* ctx, const zkw_isa_table* table); /* copies */

/* Test hooks and profiling ablations of the engine itself (nothing of the reference's surface).  Every option has an
 * environment variable of the same name (ZKW_DEBUG_FLAGS, ...) that is read ONCE, when the context is created, and
 * reported on stderr when it is set — a stray variable cannot silently change a production run, and the hot path reads
 * no environment.  zkw_ctx_set_option changes an option of a live context (what the parity tests use). */
#define ZKW_OPT_DEBUG_FLAGS 1u         /* kernel test hooks / ablations: 4 = one lane per opcode group, 1 << 24 = every light group through the variant path, 1 / 2 / 8 / 32 / 64 / 128 = store / phase ablations (WRONG results; only in a library built with -DZKW_ABLATION, ignored otherwise) */
#define ZKW_OPT_RESET_SKIP 2u          /* reset-kernel parts left out (ablation, WRONG results) */
#define ZKW_OPT_NO_INLINE_DECOMMIT 3u  /* 1 = the decommit queue is never chained inside the cycle kernel */
#define ZKW_OPT_DEBUG_SYNC 5u          /* 1 = synchronise after every cycle-kernel launch (diagnostics) */
#define ZKW_OPT_NO_GRAPH 6u            /* 1 = zkw_batch_step never captures / replays a hipGraph */
#define ZKW_OPT_WAVES_PER_GROUP 7u     /* 1 .. 8 waves per workgroup of the cycle kernel instead of the choice made per launch (0) */
#define ZKW_OPT_LANES_PER_WAVE 8u      /* overrides zkw_limits.lanes_per_wave (batches created afterwards) */
#define ZKW_OPT_PACK_BLOCKS 9u         /* workgroups of the pack kernel of deliveries created afterwards (0 = 64: the link, not the chip, bounds it) */
#define ZKW_OPT_STAGING_BUFFERS 10u     /* pinned staging buffers a batch may hold for zkw_batch_restage / zkw_batch_staging (0 = default: 4).  The heap images of
                                          a restage stay in the buffer they were handed over in for as long as something reads them — the batch's current
                                          inputs, a delivered step whose ticket is held (its memory reads travel without values and are rebuilt from the
                                          images: csrc/zkw_pack.h) — and the next restage takes another buffer of the ring; all of them taken:
                                          ZKW_ERR_LIMIT (release tickets).  1 = one buffer, overwritten by every restage: steps that ran on restaged heap
                                          images then carry the values of their memory reads on the link */
#define ZKW_OPT_READ_VALUES 11u        /* 1 = the values of memory reads always travel (the round-5 link format: A/B, tests) */
#define ZKW_OPT_LINK_FLAGS_OFF 12u     /* bits of zkw_delivered.link_flags that deliveries submitted afterwards leave out (A/B of the link format's parts, tests) */
#define ZKW_OPT_LINK_SELFCHECK 13u      /* 1 (default): the first wave a context rebuilds is packed twice — in the link format in use and in the plain one (every
                                          page, every value) — and the two rebuilds must agree; if not, the context reports it on stderr and keeps the plain
                                          format.  0 = off.  2 = behave as after a mismatch (test hook) */
int zkw_ctx_set_option(zkw_ctx* ctx, uint32_t option, uint64_t value);

int zkw_batch_create(zkw_ctx* ctx, uint32_t n_instances, const zkw_limits* limits, zkw_batch** out);
void zkw_batch_destroy(zkw_batch* batch);

/* --- staging of the initial oracle state (caller-owned inputs are copied) --- */

/* registers bytecode words (reference: SimpleMemory::populate_code memory.rs:271-284 /
 * SimpleDecommitter::populate decommitter.rs:23-28); shared by all instances */
int zkw_batch_add_code_blob(zkw_batch* batch, const zkw_u256* words, uint32_t n_words, uint32_t* blob_id);
/* known_hashes[hash] = blob (decommitter.rs:23-28) */
int zkw_batch_add_decommit_preimage(zkw_batch* batch, const zkw_u256* hash, uint32_t blob_id);
/* code_pages[page] = blob for instances [first, first+count) (memory.rs:271-284); a later call for the
 * same (instance, page) replaces the earlier one */
int zkw_batch_set_code_page(zkw_batch* batch, uint32_t first, uint32_t count, uint32_t page, uint32_t blob_id);
/* VmLocalState + callstack.inner (inner_depth entries per instance, oldest first) for
 * instances [first, first+count).  Mirrors VmState::empty_state + push_bootloader_context
 * (mod.rs:188-207, helpers.rs:289-316) having been called by the host. */
int zkw_batch_set_state(zkw_batch* batch, uint32_t first, uint32_t count, const zkw_vm_local_state* states,
                        const zkw_callstack_entry* inner, uint32_t inner_depth);
/* heap of the current (bootloader) frame (memory.rs:287-291 populate_heap) */
int zkw_batch_set_heap(zkw_batch* batch, uint32_t instance, const zkw_u256* words, uint32_t n_words);
/* SimpleMemory::polulate_bootloaders_calldata (memory.rs:293-298): the content of BOOTLOADER_CALLDATA_PAGE
 * (consts.bootloader_calldata_page).  The crate registers no indirection for that page (memory.rs:229-233, 257-261: only
 * page 0 gets one), so the VM itself cannot read it — the words come back through zkw_batch_get_page, as they do through
 * `dump_page_content` in the reference. */
int zkw_batch_set_bootloader_calldata(zkw_batch* batch, uint32_t instance, const zkw_u256* words, uint32_t n_words);
/* storage snapshot (testing/storage.rs:25-31) */
int zkw_batch_set_storage(zkw_batch* batch, uint32_t instance, const zkw_storage_slot* slots, uint32_t n_slots);
int zkw_batch_set_block_properties(zkw_batch* batch, const zkw_block_properties* props);

/* uploads everything staged so far; afterwards zkw_batch_reset restores it on device */
int zkw_batch_upload(zkw_batch* batch);
/* working state := staged initial state (device-side copy, async on `stream`; NULL = default stream) */
int zkw_batch_reset(zkw_batch* batch, void* hip_stream);
/* every instance cycles until ended or `max_cycles` more cycles (<= limits.max_cycles);
 * async on `stream` (hipStream_t; NULL = default stream) */
int zkw_batch_run(zkw_batch* batch, uint32_t max_cycles, void* hip_stream);
/* one whole step = zkw_batch_reset + zkw_batch_run(max_cycles) + zkw_batch_commit(queue_mask), enqueued on `stream`.
 * The first call runs eagerly and captures the sequence into a hipGraph; later calls with the same arguments replay
 * it with a single launch (the launch-bound regime of small batches). */
int zkw_batch_step(zkw_batch* batch, uint32_t max_cycles, uint32_t queue_mask, void* hip_stream);
/* The same step for up to 256 (ZKW_MAX_FUSED) uploaded batches of one context with FUSED launches: one reset launch,
 * one cycle-kernel launch and one set of commitment launches cover all of them (one batch per grid row).  A cycle
 * kernel launch of a small batch cannot fill the GPU (one wave per 64 instances, latency-bound), and the hardware
 * runs only a few kernels of different streams concurrently, so independent small batches are stepped together.
 * Every batch is afterwards synced / read exactly as after zkw_batch_run; kernel_ms is reported on batches[0]. */
int zkw_batches_step(zkw_batch* const* batches, uint32_t n_batches, uint32_t max_cycles, uint32_t queue_mask, void* hip_stream);
/* The three stages of zkw_batches_step as separate fused launches, for callers that pipeline groups of batches over
 * several streams (the caller orders the streams with hipStreamWaitEvent): e.g. commitments (integer-ALU bound) and
 * the restore of the next inputs on a side stream, in the shadow of the HBM-bound cycle kernel of another group. */
int zkw_batches_reset(zkw_batch* const* batches, uint32_t n_batches, void* hip_stream);
int zkw_batches_run(zkw_batch* const* batches, uint32_t n_batches, uint32_t max_cycles, void* hip_stream);
int zkw_batches_commit(zkw_batch* const* batches, uint32_t n_batches, uint32_t queue_mask, void* hip_stream);
/* zkw_batches_run for callers that will commit the queues of `queue_mask` afterwards: the cycle kernel chains what it
 * can chain while it runs (the decommit queue: one sponge permutation inside every far call that decommits) and a later
 * zkw_batches_commit / zkw_batch_commit skips those queues.  zkw_batches_step does the same implicitly. */
int zkw_batches_run_committing(zkw_batch* const* batches, uint32_t n_batches, uint32_t max_cycles, uint32_t queue_mask, void* hip_stream);
/* zkw_batches_step without its restore, for batches whose inputs are already in place: freshly uploaded, or restored by
 * a zkw_batches_reset that the caller enqueued earlier (typically right behind the previous use of the group, on a side
 * stream, so that the restore of a reused group never sits in front of its next run).  Run + commitments of one group on
 * one stream — a whole step as far as the launch geometry is concerned (helper waves).  A batch that has
 * run since its last reset continues from where it stopped, as with zkw_batches_run. */
int zkw_batches_step_prepared(zkw_batch* const* batches, uint32_t n_batches, uint32_t max_cycles, uint32_t queue_mask, void* hip_stream);
/* mean device time (ms) of the cycle-kernel launches recorded on this batch since the last call / sync — HIP event
 * pairs on the launch stream; for fused launches the pairs live on batches[0].  Waits for the last recorded launch only. */
int zkw_batch_kernel_time(zkw_batch* batch, double* mean_ms, uint32_t* n_launches);
/* waits for the run, downloads the streams and builds the per-instance views */
int zkw_batch_sync(zkw_batch* batch);
int zkw_batch_get_stats(zkw_batch* batch, zkw_run_stats* out);
int zkw_batch_get_instance_trace(zkw_batch* batch, uint32_t instance, zkw_instance_trace* out);
/* SimpleMemory::dump_page_content_as_u256_words (memory.rs:316-396) on `vm.memory` after the run (VmState.memory is a
 * public field, vm_state/mod.rs:170): words [first_word, first_word + n_words) of `page` of one instance, in the
 * reference's lookup order — code pages (populated ones and those decommitted by the run), pages with extended lifetime
 * (the bootloader calldata page; heap / aux heap pages that were returned as returndata), the stack pages and the heap /
 * aux heap pages of the frames that are live; anything else, and every word a page was never grown to, reads as zero.
 * A stack word comes back as its value (the pointer tag is dropped, :347).  One limit of the device's arena: a returndata
 * page whose receiver has returned as well (the reference never frees those) stays readable only until its arena slot is
 * reused, i.e. while the run has opened no more than limits.max_far_frames far frames.  Synchronises the batch. */
int zkw_batch_get_page(zkw_batch* batch, uint32_t instance, uint32_t page, uint32_t first_word, uint32_t n_words, zkw_u256* out);

/* The 512-byte zkw_cycle_records of instances [first, first + count) materialised on the DEVICE, for a consumer that
 * lives there: what start_new_execution_cycle(&local_state) / end_execution_cycle(&local_state) hand the tracer
 * (witness_trace/mod.rs:11-20, cycle.rs:34,413), rebuilt from the delta form the cycle kernel stores (the same records
 * zkw_batch_get_instance_trace rebuilds on the host, bit for bit).  Record k of instance i goes to
 * dst_device + ((i - first) * instance_stride + k * cycle_stride) * 512, in one of two layouts: instance-major
 * (cycle_stride = 1, instance_stride >= the cycles run since the last reset: the records of an instance are contiguous, as
 * in zkw_instance_trace.records) or cycle-major (instance_stride = 1, cycle_stride >= count: the records of a cycle are
 * contiguous — the faster one to write, a wave's 64 records of a cycle are one 32 KB run).  Both 0: instance-major with
 * limits.max_cycles records per instance.  Records beyond an instance's cycle count are left untouched.  Asynchronous on
 * `hip_stream`, ordered behind the run on the same stream (a different stream is the caller's to order).  A streaming
 * kernel: 512 bytes written per VM cycle. */
int zkw_batch_expand_records(zkw_batch* batch, uint32_t first, uint32_t count, void* dst_device, uint64_t instance_stride, uint64_t cycle_stride, void* hip_stream);
/* The same for every instance of up to 256 batches of one context in fused launches (what a device-side consumer of a
 * zkw_batches_step calls): dst_device[b] receives batch b, laid out as above with first = 0.  One batch is 64 waves of a
 * sequential chain each, so a lone batch is expanded in chunks of cycles (every chunk replays the cycles in front of it
 * silently); a fused group has waves enough to run at the HBM write rate unchunked. */
int zkw_batches_expand_records(zkw_batch* const* batches, uint32_t n_batches, void* const* dst_device, uint64_t instance_stride, uint64_t cycle_stride, void* hip_stream);

/* --- queue commitments (the build's own sponge spec) --- */
#define ZKW_QUEUE_MEMORY 0
#define ZKW_QUEUE_LOG 1
#define ZKW_QUEUE_DECOMMIT 2
#define ZKW_QUEUE_COUNT 3
/* enqueues the commitment kernels for the queues in `queue_mask` (bit q = ZKW_QUEUE_q) on `stream`, after
 * the run(s) since the last reset; results stay in device memory (zkw_batch_commitments_device_ptr) */
int zkw_batch_commit(zkw_batch* batch, uint32_t queue_mask, void* hip_stream);
/* digests[instance][queue] : 4 x u64 Goldilocks elements each, computed on device from the
 * streams of the last run; `out` holds n_instances * ZKW_QUEUE_COUNT * 4 u64 */
int zkw_batch_get_commitments(zkw_batch* batch, uint64_t* out);
/* async device-to-device copy of the digests (n_instances * ZKW_QUEUE_COUNT * 4 u64) into a caller buffer,
 * e.g. the send buffer of the RCCL all-gather */
int zkw_batch_copy_commitments(zkw_batch* batch, void* dst_device, void* hip_stream);
/* device pointer (n_instances * ZKW_QUEUE_COUNT * 4 u64) for the RCCL all-gather */
int zkw_batch_commitments_device_ptr(zkw_batch* batch, void** dptr, uint64_t* n_bytes);

/* ---- final exchange across the ranks of a job (SURVEY §8b `zkw_reduce_commitments`, §8e) ----
 * VM instances shard across GPUs with no data-path collective; the only exchange is the final one: an all-gather of
 * the per-instance queue digests and an all-reduce (sum) of the run counters.  One process per GPU.  Two transports:
 *   RCCL over xGMI   rank 0 calls zkw_comm_get_unique_id, hands the 128 bytes to the other ranks over any side
 *                    channel (a file, a socket, torch.distributed), then every rank calls zkw_comm_create_rccl;
 *                    librccl is loaded at that moment (dlopen), libzkw.so itself does not link it;
 *   external         the caller supplies the two collectives over HOST buffers (MPI, gloo, ...): same entry point,
 *                    same packing, the exchange itself runs through the callbacks. */
typedef struct zkw_comm zkw_comm;
typedef struct zkw_comm_id {
  uint8_t bytes[128]; /* ncclUniqueId */
} zkw_comm_id;
/* Side-effect free: ZKW_OK when librccl can be loaded and has the symbols the RCCL transport uses (dlopen + dlsym only).
 * What every rank checks BEFORE rank 0 makes the id: ncclGetUniqueId starts a bootstrap root (a listening socket and a
 * thread), which only the rank whose id is used should own. */
int zkw_comm_probe(void);
int zkw_comm_get_unique_id(zkw_comm_id* out);
int zkw_comm_create_rccl(zkw_ctx* ctx, int rank, int world, const zkw_comm_id* id, zkw_comm** out);
/* recv = world x bytes_per_rank, rank order; return 0 on success */
typedef int (*zkw_allgather_fn)(void* user, const void* send, void* recv, uint64_t bytes_per_rank);
typedef int (*zkw_allreduce_sum_u64_fn)(void* user, uint64_t* inout, uint32_t count);
int zkw_comm_create_external(zkw_ctx* ctx, int rank, int world, zkw_allgather_fn allgather, zkw_allreduce_sum_u64_fn allreduce_sum,
                             void* user, zkw_comm** out);
void zkw_comm_destroy(zkw_comm* comm);
/* COLLECTIVE: every rank announces the n_instances of the batches it will reduce from now on (waits for the stream). */
int zkw_comm_exchange_sizes(zkw_comm* comm, uint32_t n_instances, void* hip_stream);
/* The final exchange for `n_batches` committed batches of this rank (zkw_batch_commit / zkw_batches_commit with at
 * least the queues of `queue_mask`, enqueued earlier on `hip_stream` or ordered before it by the caller).  Every rank
 * passes the same n_batches and queue_mask; ranks may own different numbers of instances per batch (ragged shards: all
 * batches of one rank have the same n_instances; rows are padded to the largest rank, sizes[] tells them apart).
 *   gathered   [world][n_batches][n_max][popcount(queue_mask)][4] u64, queues in ascending order — DEVICE memory for an
 *              RCCL communicator (the all-gather is enqueued on `hip_stream`, asynchronously), HOST memory for an
 *              external one (the call then waits for the stream).  NULL: no digest exchange.
 *   n_max_out  optional: rows per rank in `gathered` (= max over the ranks of n_instances)
 *   sizes_out  optional [world]: n_instances of every rank
 *   total      optional: the run counters of these batches summed over all ranks (cycles, queries, aux events, ended /
 *              failed instances, register deltas; kernel_ms = max over the ranks).  Waits for the stream (the counters
 *              come from the finished runs).  NULL: no counter exchange.
 * The FIRST call of a communicator exchanges the shard sizes (a collective, and it waits for it); later calls with
 * gathered != NULL and total == NULL are fully asynchronous — also from several streams at once (one send buffer per
 * stream).  A rank whose n_instances changes afterwards gets ZKW_ERR_INVALID until EVERY rank has called
 * zkw_comm_exchange_sizes: a rank-local decision to exchange again would pair a size exchange on one rank with a digest
 * all-gather on another. */
int zkw_reduce_commitments(zkw_comm* comm, zkw_batch* const* batches, uint32_t n_batches, uint32_t queue_mask, void* gathered,
                           uint32_t* n_max_out, uint32_t* sizes_out, zkw_run_stats* total, void* hip_stream);

/* ---- final net states: what the reference's get_final_net_states (testing/mod.rs:42-71) returns ----
 * The device nets the log stream of every instance against its frame events exactly like
 * InMemoryStorage::finish_frame / flatten_and_net_history (testing/storage.rs:34-76,144-186) and
 * InMemoryEventSink::finish_frame / flatten (reference_impls/event_sink.rs:66-131,160-176) do on the host: the access
 * history is the chronological list of queries with, at the finish of every panicked frame, that frame's pending
 * rollback entries appended in reverse order (rollback flag set); net events / L1 messages are the event queries that
 * were never rolled back, in timestamp order.  Instances that are still running are netted as if their open frames
 * were kept.  Failed instances (status >= ZKW_STATUS_UNKNOWN_CODE_HASH) have no net state. */
typedef struct zkw_event_message { /* EventMessage, reference_impls/event_sink.rs:7-14 */
  uint8_t shard_id;
  uint8_t is_first;            /* = LogQuery.is_service */
  uint16_t tx_number_in_block;
  uint8_t address[20];
  zkw_u256 key;
  zkw_u256 value;              /* = LogQuery.written_value */
} zkw_event_message;           /* 88 B */

typedef struct zkw_net_state {
  uint32_t n_storage_history, n_event_history, n_events, n_l1_messages, n_final_storage, reserved0;
  const zkw_log_query* storage_history; /* full_storage_access_history; lane / seq / kind are zero, ZKW_LQ_ROLLBACK marks rollback entries, reads carry written_value = 0 (what the Storage saw: log.rs:175, far_call.rs:139) */
  const zkw_log_query* event_history;   /* events_log_history, same conventions */
  const zkw_event_message* events;      /* aux_byte == event_aux_byte */
  const zkw_event_message* l1_messages; /* every other surviving event-sink query (event_sink.rs:124-128) */
  const zkw_storage_slot* final_storage; /* final_storage_state: populated or written slots, sorted by (shard, address, key) */
} zkw_net_state;

/* enqueues the netting pass over the finished run on `stream` (bucket passes over the log / aux streams + one
 * sequential walk per instance, one instance per lane) */
int zkw_batch_net_states(zkw_batch* batch, void* hip_stream);
/* downloads and materialises the net state of one instance (library-owned arrays, valid until the next call for this
 * batch); runs zkw_batch_net_states first if it has not been run since the last run.  ZKW_ERR_LIMIT if the instance
 * overflowed its per-instance index capacity (limits.max_log_queries / max_aux_events), ZKW_ERR_INVALID for a failed
 * instance. */
int zkw_batch_get_net_state(zkw_batch* batch, uint32_t instance, zkw_net_state* out);

/* ---- delivery to the host (SURVEY §8d ii "run + record download"; the drop-in the north star names is a HOST tracer) ----
 * All ten VmWitnessTracer callbacks (reference src/witness_trace/mod.rs:11-72) run on the CPU, so the witness of every step
 * has to cross PCIe.  A zkw_delivery owns a persistent ring of pinned host slots (allocated once), a stream of its own and a
 * pool of host threads.  zkw_delivery_submit enqueues, behind the run of a group of batches, ONE pack kernel that writes the
 * used extents of everything the step produced — every wave of every batch: directory, record tails, register deltas, the
 * three query streams, the final scalars / register files / callstack entries — as one contiguous block straight into a
 * slot of the ring (the kernel's stores are the transfer; the block is a denser encoding than the device streams:
 * era-zk_evm_amd/csrc/zkw_pack.h).  Step k's delivery runs beside step k + 1's cycle kernel when the caller alternates two
 * groups of batches.  The host side of a delivered step:
 *   zkw_delivery_get_instance_trace   the zkw_instance_trace of one instance, rebuilt from the ring (what
 *                                     zkw_batch_get_instance_trace returns, bit for bit, without touching the device)
 *   zkw_delivery_replay               every (instance, cycle) of the step handed to a callback on the pool's threads, waves in
 *                                     parallel, with a pointer to the instance's LIVE 512-byte snapshot — the contract of
 *                                     start_new_execution_cycle(&local_state) / end_execution_cycle(&local_state)
 *                                     (witness_trace/mod.rs:11-20): a reference to the state, not a copy per cycle — and the
 *                                     cycle's queries in emission order
 * Tickets count submissions (0, 1, 2, ...); ticket t lives in slot t % n_slots until it is released. */
typedef struct zkw_delivery zkw_delivery;
typedef struct zkw_delivered {
  uint64_t bytes;     /* bytes of the block that crossed the link */
  double pack_ms;     /* device time of the pack kernel (HIP events on the delivery's stream) */
  uint32_t n_batches, n_waves;
  uint32_t overflow;  /* != 0: the step did not fit the slot (slot_bytes too small): nothing of it can be read */
  uint32_t link_flags; /* what did NOT travel because the rebuild derives it (link format, csrc/zkw_pack.h): 1 = the values of memory
                          reads, 2 = the pages of the VM's own stack / heap / code queries, 4 = the event counts of the record tails, 8 = the zero upper
                          bytes of register deltas, 16 = pc / sp / ergs / pointer bitmap of a record tail where the previous tail predicts them */
} zkw_delivered;
/* n_slots >= 1 slots of slot_bytes each (hipHostMalloc, once); host_threads >= 1 worker threads for the replay / rebuild */
int zkw_delivery_create(zkw_ctx* ctx, uint32_t n_slots, uint64_t slot_bytes, uint32_t host_threads, zkw_delivery** out);
void zkw_delivery_destroy(zkw_delivery* d);
/* Upper bound of the block a step of these batches can produce (every stream at its capacity): what slot_bytes must cover
 * in the worst case; a typical step uses a fraction (zkw_delivered.bytes). */
int zkw_delivery_slot_bytes(zkw_batch* const* batches, uint32_t n_batches, uint64_t* worst_case);
/* Delivers the step the batches have just run on `run_stream` (asynchronous: an event on run_stream orders the pack kernel
 * behind the run; nothing waits on the host).  ZKW_ERR_LIMIT while the slot of the new ticket is still held. */
int zkw_delivery_submit(zkw_delivery* d, zkw_batch* const* batches, uint32_t n_batches, void* run_stream, uint32_t* ticket);
/* Makes `hip_stream` wait (on the device) until the pack kernel of `ticket` has read the batches' streams: what the next
 * reset / restage / run of those batches has to be ordered behind. */
int zkw_delivery_order_after(zkw_delivery* d, uint32_t ticket, void* hip_stream);
/* Blocks the host until the block of `ticket` is in the ring. */
int zkw_delivery_wait(zkw_delivery* d, uint32_t ticket, zkw_delivered* info);
int zkw_delivery_get_instance_trace(zkw_delivery* d, uint32_t ticket, uint32_t batch_index, uint32_t instance, zkw_instance_trace* out);
/* `thread` = index of the pool thread that calls (0 .. host_threads - 1): calls of one instance come in cycle order from one
 * thread, different instances from different threads at once; the pointers are valid during the call only */
typedef void (*zkw_cycle_fn)(void* user, uint32_t thread, uint32_t batch_index, uint32_t instance, uint32_t cycle, const zkw_cycle_record* state_after,
                             const zkw_mem_query* mem, uint32_t n_mem, const zkw_log_query* log, uint32_t n_log, const zkw_aux_event* aux, uint32_t n_aux);
/* fn == NULL: a built-in consumer that reads every byte handed over and folds it into `checksum`: the sum, mod 2^64, over all
 * records and queries of sum_j (u64[j] ^ K * (j + 1 + w0)) with K = 0x9E3779B97F4A7C15 and w0 = 1 / 3 / 5 / 7 for cycle records /
 * memory / log / aux records — order-independent, so that a test can compare it with the same fold over
 * zkw_delivery_get_instance_trace.  n_cycles / checksum may be NULL. */
int zkw_delivery_replay(zkw_delivery* d, uint32_t ticket, zkw_cycle_fn fn, void* user, uint64_t* n_cycles, uint64_t* checksum);
int zkw_delivery_release(zkw_delivery* d, uint32_t ticket);

/* ---- fresh inputs for an uploaded batch (the input side of a pipelined caller) ----
 * New VmLocalStates (registers, flags, pc / sp / ergs ... of the current frame — VmState::empty_state +
 * push_bootloader_context with other values, reference src/vm_state/mod.rs:188-207, helpers.rs:289-316) and new heap
 * images (SimpleMemory::populate_heap, memory.rs:287-291) for EVERY instance of a batch that keeps its geometry: same
 * limits, code, decommit preimages, storage snapshot, callstack depth and heap-image length as uploaded.  The host formats
 * the images into pinned staging memory of the batch (allocated on first use), one H2D copy per image is enqueued on
 * `hip_stream` and behind them the device-side restore — asynchronous: the call returns when the copies are enqueued, and a
 * caller restages group B on a side stream while group A runs.  states [n_instances]; heap_words [n_instances][n_heap_words]
 * with n_heap_words == the uploaded image length (or NULL / 0: heaps unchanged).  The staged inputs of the batch are updated:
 * traces rebuilt afterwards replay onto the new initial states.
 * The interleaved device layouts are produced ON the device (zkw_restage_kernel): the host only copies the caller's arrays
 * into pinned memory — or not even that: zkw_batch_staging hands out the pinned buffers themselves (states [n_instances],
 * heap_words [n_instances][*n_heap_words]), a caller that builds its inputs there passes those pointers to zkw_batch_restage
 * and nothing is copied on the host.  The pointers are good for ONE restage: a batch holds a small ring of staging buffers
 * (ZKW_OPT_STAGING_BUFFERS), the heap images of a restage stay in their buffer while a delivered step still rebuilds its memory
 * reads from them, and every zkw_batch_staging hands out a buffer nobody reads (waiting for the H2D copies that last used it).  After a restage with heap images the library no longer holds the batch's heaps on the host: another
 * zkw_batch_upload needs zkw_batch_set_heap again.
 * The uploaded image length is the LONGEST heap any instance was given (zkw_batch_set_heap); a restaged image has that length
 * for every instance, whatever the instance itself had uploaded — all of its words are readable afterwards.
 * Delivery tickets are self-contained: a ticket submitted before the restage (the order a pipeline wants: submit,
 * zkw_delivery_order_after on the side stream, restage) keeps the inputs its own step ran on, so its traces / replay are
 * unchanged by the restage — and by a later zkw_batch_upload or zkw_batch_destroy of the batch. */
int zkw_batch_staging(zkw_batch* batch, zkw_vm_local_state** states, zkw_u256** heap_words, uint32_t* n_heap_words);
int zkw_batch_restage(zkw_batch* batch, const zkw_vm_local_state* states, const zkw_u256* heap_words, uint32_t n_heap_words, void* hip_stream);

/* Pulls everything the last run produced (record tails, register deltas, the three query streams, the directory; used
 * extents only) over PCIe into pinned staging memory and reports the volume and the transfer time: the cost a host-side
 * consumer of the whole trace would pay (PCIe-inclusive rate).  The data is discarded. */
int zkw_batch_download_all(zkw_batch* batch, uint64_t* n_bytes, double* ms);

/* BLAKE2s-256 (RFC 7693: unkeyed, 32-byte digests, no salt / personalisation) of n_messages byte strings on the GPU,
 * one message per lane.  Replaces: `Blake2s256::digest(bytes)` of the `blake2` crate as callers reach it through the
 * reference's re-export `zk_evm::blake2` (/root/reference/src/lib.rs:21 — the reference itself has no call site).
 * Message i is data[offsets[i] .. offsets[i + 1]) (offsets has n_messages + 1 non-decreasing entries; an empty message
 * is allowed); digests receives n_messages x 32 bytes in message order.
 * zkw_blake2s256: host buffers, synchronous (staged through device buffers the context keeps).
 * zkw_blake2s256_device: the buffers are device memory and the kernel is enqueued on `hip_stream` without a
 * synchronisation; d_data must be 4-byte aligned and its allocation must cover total_bytes rounded up to 4,
 * d_offsets 8-byte and d_digests 16-byte aligned; the offsets are trusted. */
int zkw_blake2s256(zkw_ctx* ctx, const uint8_t* data, const uint64_t* offsets, uint32_t n_messages, uint8_t* digests);
int zkw_blake2s256_device(zkw_ctx* ctx, const void* d_data, uint64_t total_bytes, const uint64_t* d_offsets, uint32_t n_messages, void* d_digests,
                          void* hip_stream);

/* sizeof() of the ABI structs as compiled into the library (binding self-check) */
uint32_t zkw_abi_sizeof(uint32_t which);

#ifdef __cplusplus
}
#endif
#endif /* ZKW_H */
