// Default ISA table: the build's recollection of zkevm_opcode_defs v1.4.1.
//
// The reference reads the decode table, prices and scalar constants from the un-vendored
// crate `zkevm_opcode_defs` (branch v1.4.1, reference Cargo.toml:15; call sites cycle.rs:135-148,
// 174-178, 341, 375-390).  Its source is not available offline, so NOTHING here is verified
// (SURVEY.md Appendix B): the engine is table-driven and a Rust shim overwrites this table
// with the real OPCODES_TABLE / OPCODES_PRICES through zkw_ctx_set_isa.
// The table is synthesised the way the crate does it — every opcode variant x src0 addressing
// mode x dst0 addressing mode x flag combination gets one 11-bit index — so the variant count
// (1092 + invalid padding) and the structure match; the numbering is the build's own.
//
// Host-only code: no HIP calls, usable without a GPU.
#include <cstring>

#include "../../include/zkw.h"

namespace {

const uint32_t VM_CYCLE = 4, RAM_PERM = 1, LOG_DEMUX = 1, STORAGE_SORTER = 2, EVENTS_SORTER = 1, DECOMMIT_SORTER = 1;
const uint32_t RICH = VM_CYCLE + 4 * RAM_PERM;     // RICH_ADDRESSING_OPCODE_ERGS
const uint32_t AVERAGE = VM_CYCLE + 2 * RAM_PERM;  // AVERAGE_OPCODE_ERGS
const uint32_t CALL_LIKE = 20, STORAGE_READ_IO = 150, STORAGE_WRITE_IO = 250, EVENT_IO = 25, L1_MESSAGE_MIN_COST = 156250;

struct Family {
  uint8_t opcode, variant;
  uint8_t src_full;  // 1: src0 is Operand::Full (6 modes); 2: RegOrImm (REG, IMM); 0: register only
  uint8_t dst_full;  // 1: dst0 is Operand::Full (4 modes); 0: register only
  uint8_t n_flags;
  uint8_t props;     // KERNEL_ONLY / STATIC_OK / SRC0_PTR_OK
  uint8_t swap_flag; // 0: none, else 1 + index of the flag that swaps operands
  uint32_t price;
};

const uint8_t K = ZKW_PROP_KERNEL_ONLY, S = ZKW_PROP_STATIC_OK, P = ZKW_PROP_SRC0_PTR_OK;

const Family FAMILIES[] = {
    {ZKW_OP_NOP, 0, 1, 1, 0, S, 0, RICH},
    {ZKW_OP_ADD, 0, 1, 1, 1, S, 0, RICH},
    {ZKW_OP_SUB, 0, 1, 1, 2, S, 2, RICH},
    {ZKW_OP_MUL, 0, 1, 1, 1, S, 0, RICH},
    {ZKW_OP_DIV, 0, 1, 1, 2, S, 2, RICH},
    {ZKW_OP_JUMP, 0, 1, 0, 0, S, 0, RICH},
    {ZKW_OP_CONTEXT, ZKW_CTX_THIS, 0, 0, 0, S, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_CALLER, 0, 0, 0, S, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_CODE_ADDRESS, 0, 0, 0, S, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_META, 0, 0, 0, S, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_ERGS_LEFT, 0, 0, 0, S, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_SP, 0, 0, 0, S, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_GET_CONTEXT_U128, 0, 0, 0, S, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_SET_CONTEXT_U128, 0, 0, 0, K, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_SET_ERGS_PER_PUBDATA, 0, 0, 0, K, 0, AVERAGE},
    {ZKW_OP_CONTEXT, ZKW_CTX_INC_TX_NUMBER, 0, 0, 0, K, 0, AVERAGE},
    {ZKW_OP_SHIFT, ZKW_SHIFT_SHL, 1, 1, 2, S, 2, RICH},
    {ZKW_OP_SHIFT, ZKW_SHIFT_SHR, 1, 1, 2, S, 2, RICH},
    {ZKW_OP_SHIFT, ZKW_SHIFT_ROL, 1, 1, 2, S, 2, RICH},
    {ZKW_OP_SHIFT, ZKW_SHIFT_ROR, 1, 1, 2, S, 2, RICH},
    {ZKW_OP_BINOP, ZKW_BINOP_XOR, 1, 1, 1, S, 0, RICH},
    {ZKW_OP_BINOP, ZKW_BINOP_AND, 1, 1, 1, S, 0, RICH},
    {ZKW_OP_BINOP, ZKW_BINOP_OR, 1, 1, 1, S, 0, RICH},
    {ZKW_OP_PTR, ZKW_PTR_ADD, 1, 1, 1, (uint8_t)(S | P), 1, RICH},
    {ZKW_OP_PTR, ZKW_PTR_SUB, 1, 1, 1, (uint8_t)(S | P), 1, RICH},
    {ZKW_OP_PTR, ZKW_PTR_PACK, 1, 1, 1, (uint8_t)(S | P), 1, RICH},
    {ZKW_OP_PTR, ZKW_PTR_SHRINK, 1, 1, 1, (uint8_t)(S | P), 1, RICH},
    {ZKW_OP_NEAR_CALL, 0, 0, 0, 0, S, 0, AVERAGE + CALL_LIKE},
    {ZKW_OP_LOG, ZKW_LOG_STORAGE_READ, 0, 0, 0, S, 0, VM_CYCLE + RAM_PERM + LOG_DEMUX + STORAGE_SORTER + STORAGE_READ_IO},
    {ZKW_OP_LOG, ZKW_LOG_STORAGE_WRITE, 0, 0, 0, 0, 0, 2 * VM_CYCLE + RAM_PERM + 2 * LOG_DEMUX + 2 * STORAGE_SORTER + STORAGE_WRITE_IO},
    {ZKW_OP_LOG, ZKW_LOG_TO_L1, 0, 0, 1, K, 0, L1_MESSAGE_MIN_COST},
    {ZKW_OP_LOG, ZKW_LOG_EVENT, 0, 0, 1, K, 0, VM_CYCLE + RAM_PERM + 2 * LOG_DEMUX + 2 * EVENTS_SORTER + EVENT_IO},
    {ZKW_OP_LOG, ZKW_LOG_PRECOMPILE, 0, 0, 0, (uint8_t)(K | S), 0, VM_CYCLE + RAM_PERM + LOG_DEMUX},
    {ZKW_OP_FAR_CALL, ZKW_FAR_NORMAL, 0, 0, 2, (uint8_t)(S | P), 0, 2 * VM_CYCLE + RAM_PERM + STORAGE_READ_IO + CALL_LIKE + STORAGE_SORTER + DECOMMIT_SORTER},
    {ZKW_OP_FAR_CALL, ZKW_FAR_DELEGATE, 0, 0, 2, (uint8_t)(S | P), 0, 2 * VM_CYCLE + RAM_PERM + STORAGE_READ_IO + CALL_LIKE + STORAGE_SORTER + DECOMMIT_SORTER},
    {ZKW_OP_FAR_CALL, ZKW_FAR_MIMIC, 0, 0, 2, (uint8_t)(K | S | P), 0, 2 * VM_CYCLE + RAM_PERM + STORAGE_READ_IO + CALL_LIKE + STORAGE_SORTER + DECOMMIT_SORTER},
    {ZKW_OP_RET, ZKW_RET_OK, 0, 0, 1, (uint8_t)(S | P), 0, AVERAGE},
    {ZKW_OP_RET, ZKW_RET_REVERT, 0, 0, 1, (uint8_t)(S | P), 0, AVERAGE},
    {ZKW_OP_RET, ZKW_RET_PANIC, 0, 0, 1, (uint8_t)(S | P), 0, AVERAGE},
    {ZKW_OP_UMA, ZKW_UMA_HEAP_READ, 2, 0, 1, S, 0, VM_CYCLE + 3 * RAM_PERM},
    {ZKW_OP_UMA, ZKW_UMA_HEAP_WRITE, 2, 0, 1, S, 0, VM_CYCLE + 5 * RAM_PERM},
    {ZKW_OP_UMA, ZKW_UMA_AUX_READ, 2, 0, 1, S, 0, VM_CYCLE + 3 * RAM_PERM},
    {ZKW_OP_UMA, ZKW_UMA_AUX_WRITE, 2, 0, 1, S, 0, VM_CYCLE + 5 * RAM_PERM},
    {ZKW_OP_UMA, ZKW_UMA_FAT_PTR_READ, 0, 0, 1, (uint8_t)(S | P), 0, VM_CYCLE + 3 * RAM_PERM},
};

const uint8_t SRC_FULL[] = {ZKW_MODE_REG, ZKW_MODE_STACK_PP, ZKW_MODE_STACK_OFF, ZKW_MODE_STACK_ABS, ZKW_MODE_IMM, ZKW_MODE_CODE};
const uint8_t SRC_REG_OR_IMM[] = {ZKW_MODE_REG, ZKW_MODE_IMM};
const uint8_t DST_FULL[] = {ZKW_MODE_REG, ZKW_MODE_STACK_PP, ZKW_MODE_STACK_OFF, ZKW_MODE_STACK_ABS};
const uint8_t ONLY_REG[] = {ZKW_MODE_REG};

}  // namespace

extern "C" {

int zkw_isa_default(zkw_isa_table* out) {
  if (!out) return ZKW_ERR_INVALID;
  std::memset(out, 0, sizeof *out);
  zkw_isa_entry invalid;
  std::memset(&invalid, 0, sizeof invalid);
  invalid.opcode = ZKW_OP_INVALID;
  invalid.props = ZKW_PROP_EXPLICIT_PANIC | ZKW_PROP_STATIC_OK;
  invalid.price = 0xffffffffu;  // INVALID_OPCODE_ERGS
  for (int i = 0; i < ZKW_ISA_TABLE_SIZE; i++) out->entries[i] = invalid;
  uint32_t idx = 1;  // index 0 stays Invalid (an all-zero code word decodes to it)
  for (const Family& f : FAMILIES) {
    const uint8_t* srcs = f.src_full == 1 ? SRC_FULL : (f.src_full == 2 ? SRC_REG_OR_IMM : ONLY_REG);
    int n_src = f.src_full == 1 ? 6 : (f.src_full == 2 ? 2 : 1);
    const uint8_t* dsts = f.dst_full ? DST_FULL : ONLY_REG;
    int n_dst = f.dst_full ? 4 : 1;
    for (int s = 0; s < n_src; s++)
      for (int d = 0; d < n_dst; d++)
        for (uint32_t fl = 0; fl < (1u << f.n_flags); fl++) {
          if (idx >= ZKW_ISA_TABLE_SIZE) return ZKW_ERR_LIMIT;
          zkw_isa_entry& e = out->entries[idx++];
          e.opcode = f.opcode;
          e.variant = f.variant;
          e.src0_mode = srcs[s];
          e.dst0_mode = dsts[d];
          e.flags = (uint8_t)fl;
          e.props = f.props;
          if (f.swap_flag && (fl >> (f.swap_flag - 1)) & 1) e.props |= ZKW_PROP_SWAP;
          e.price = f.price;
        }
  }
  zkw_isa_consts& c = out->consts;
  c.nop_variant_idx = (uint32_t)zkw_isa_find(out, ZKW_OP_NOP, 0, ZKW_MODE_REG, ZKW_MODE_REG, 0);
  c.panic_variant_idx = (uint32_t)zkw_isa_find(out, ZKW_OP_RET, ZKW_RET_PANIC, ZKW_MODE_REG, ZKW_MODE_REG, 0);
  c.nop_encoding = zkw_isa_encode(c.nop_variant_idx, 0, 0, 0, 0, 0, 0, 0);
  c.exception_revert_encoding = zkw_isa_encode(c.panic_variant_idx, 0, 0, 0, 0, 0, 0, 0);
  c.clip_mode = 0;
  c.time_delta_per_cycle = 4;
  c.new_memory_pages_per_far_call = 8;
  c.vm_max_stack_depth = 0xffffffffu / CALL_LIKE + 80;  // VM_INITIAL_FRAME_ERGS / CALL_LIKE_ERGS_COST + 80
  c.initial_sp_on_far_call = 0;
  c.new_frame_memory_stipend = 1u << 12;
  c.memory_growth_ergs_per_byte = 1;
  c.ergs_per_code_word_decommittment = 4;
  c.initial_storage_write_pubdata_bytes = 64;
  c.l1_message_pubdata_bytes = 1 + 1 + 2 + 20 + 32 + 32;
  c.max_offset_to_deref_low = 0xffffffffu - 32;
  c.deployer_address_low = 0x8002;
  c.keccak_precompile_address = 0x8010;
  c.sha256_precompile_address = 0x02;
  c.ecrecover_precompile_address = 0x01;
  c.storage_aux_byte = 0;
  c.event_aux_byte = 1;
  c.l1_message_aux_byte = 2;
  c.precompile_aux_byte = 3;
  c.bootloader_calldata_page = 3;  // recalled from zkevm_opcode_defs (UNVERIFIED; SURVEY App. B lists it as unknown): a table constant for that reason
  c.call_regs = 0u | (1u << 8) | (14u << 16);
  c.call_ranges = 2u | (12u << 8) | (12u << 16) | (14u << 24);
  c.ret_regs = 0u | (1u << 8) | (2u << 16) | (3u << 24);
  c.forwarding_codes = 0u | (1u << 8) | (2u << 16);
  c.unmapped_page = 0;
  c.reserved0 = 0;
  c.max_offset_for_add_sub = 1ull << 32;
  // Always, Gt, Lt, Eq, Ge, Le, Ne, GtOrLt over (lt | eq << 1 | gt << 2)
  c.condition_lut = 0xffull | (0xf0ull << 8) | (0xaaull << 16) | (0xccull << 24) | (0xfcull << 32) | (0xeeull << 40) | (0x33ull << 48) | (0xfaull << 56);
  return ZKW_OK;
}

uint64_t zkw_isa_encode(uint32_t variant_idx, uint32_t condition, uint32_t src0, uint32_t src1, uint32_t dst0, uint32_t dst1, uint32_t imm0,
                        uint32_t imm1) {
  return (uint64_t)(variant_idx & 0x7ff) | ((uint64_t)(condition & 7) << 13) | ((uint64_t)(src0 & 15) << 16) | ((uint64_t)(src1 & 15) << 20) |
         ((uint64_t)(dst0 & 15) << 24) | ((uint64_t)(dst1 & 15) << 28) | ((uint64_t)(imm0 & 0xffff) << 32) | ((uint64_t)(imm1 & 0xffff) << 48);
}

int32_t zkw_isa_find(const zkw_isa_table* t, uint32_t opcode, uint32_t variant, uint32_t src0_mode, uint32_t dst0_mode, uint32_t flags) {
  for (int i = 0; i < ZKW_ISA_TABLE_SIZE; i++) {
    const zkw_isa_entry& e = t->entries[i];
    if (e.opcode == opcode && e.variant == variant && e.src0_mode == src0_mode && e.dst0_mode == dst0_mode && e.flags == flags) return i;
  }
  return -1;
}

}  // extern "C"
