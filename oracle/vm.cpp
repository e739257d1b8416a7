// ORACLE — TEST INFRASTRUCTURE ONLY (see vm.hpp header).
// Restatement of reference src/vm_state/cycle.rs, mem_ops.rs and src/opcodes/execution/*.rs.
#include "vm.hpp"
#include "secp256k1.hpp"

namespace zko {

// ---------------------------------------------------------------------------------------
// conversions to the C structs of include/zkw.h
// ---------------------------------------------------------------------------------------
void entry_to_c(const CallStackEntry& e, zkw_callstack_entry* o) {
  std::memset(o, 0, sizeof *o);
  std::memcpy(o->this_address, e.this_address.b, 20);
  std::memcpy(o->msg_sender, e.msg_sender.b, 20);
  std::memcpy(o->code_address, e.code_address.b, 20);
  o->base_memory_page = e.base_memory_page; o->code_page = e.code_page;
  o->sp = e.sp; o->pc = e.pc; o->exception_handler_location = e.exception_handler_location;
  o->is_static = e.is_static; o->is_local_frame = e.is_local_frame;
  o->ergs_remaining = e.ergs_remaining;
  o->this_shard_id = e.this_shard_id; o->caller_shard_id = e.caller_shard_id; o->code_shard_id = e.code_shard_id;
  o->context_u128_value[0] = e.context_u128_value[0]; o->context_u128_value[1] = e.context_u128_value[1];
  o->heap_bound = e.heap_bound; o->aux_heap_bound = e.aux_heap_bound;
}
void entry_from_c(const zkw_callstack_entry& c, CallStackEntry* e) {
  std::memcpy(e->this_address.b, c.this_address, 20);
  std::memcpy(e->msg_sender.b, c.msg_sender, 20);
  std::memcpy(e->code_address.b, c.code_address, 20);
  e->base_memory_page = c.base_memory_page; e->code_page = c.code_page;
  e->sp = c.sp; e->pc = c.pc; e->exception_handler_location = c.exception_handler_location;
  e->is_static = c.is_static != 0; e->is_local_frame = c.is_local_frame != 0;
  e->ergs_remaining = c.ergs_remaining;
  e->this_shard_id = c.this_shard_id; e->caller_shard_id = c.caller_shard_id; e->code_shard_id = c.code_shard_id;
  e->context_u128_value[0] = c.context_u128_value[0]; e->context_u128_value[1] = c.context_u128_value[1];
  e->heap_bound = c.heap_bound; e->aux_heap_bound = c.aux_heap_bound;
}
void state_to_c(const VmLocalState& s, zkw_vm_local_state* o) {
  std::memset(o, 0, sizeof *o);
  std::memcpy(o->previous_code_word.l, s.previous_code_word.l, 32);
  uint16_t bitmap = 0;
  for (int i = 0; i < ZKW_REGISTERS_COUNT; i++) {
    std::memcpy(o->registers[i].l, s.registers[i].value.l, 32);
    if (s.registers[i].is_pointer) bitmap |= (uint16_t)(1u << i);
  }
  o->register_ptr_bitmap = bitmap;
  o->flags = (uint8_t)((s.flags.overflow_or_less_than_flag ? 1 : 0) | (s.flags.equality_flag ? 2 : 0) | (s.flags.greater_than_flag ? 4 : 0));
  o->pending_exception = s.pending_exception;
  o->previous_code_memory_page = s.previous_code_memory_page;
  o->timestamp = s.timestamp; o->monotonic_cycle_counter = s.monotonic_cycle_counter;
  o->spent_pubdata_counter = s.spent_pubdata_counter; o->memory_page_counter = s.memory_page_counter;
  o->absolute_execution_step = s.absolute_execution_step; o->current_ergs_per_pubdata_byte = s.current_ergs_per_pubdata_byte;
  o->tx_number_in_block = s.tx_number_in_block; o->previous_super_pc = s.previous_super_pc;
  o->callstack_depth = (uint32_t)s.callstack.depth();
  o->context_u128_register[0] = s.context_u128_register[0]; o->context_u128_register[1] = s.context_u128_register[1];
  entry_to_c(s.callstack.current, &o->current);
}

// ---------------------------------------------------------------------------------------
// mem_ops.rs:14-125 compute_addresses_and_select_operands
// returns true when a MemoryLocation was produced
// ---------------------------------------------------------------------------------------
bool Vm::compute_addresses(uint16_t& sp, uint8_t reg_idx, uint16_t imm, uint8_t mode, bool is_write, PrimitiveValue* reg_value, MemoryLocation* loc) {
  PrimitiveValue pv = select_register_value(reg_idx);
  *reg_value = pv;
  uint16_t reg_low = clip16(pv.value.low_u64());        // :34
  uint16_t vaddr = (uint16_t)(reg_low + imm);           // :35 wrapping_add
  const CallStackEntry& ctx = local_state.callstack.current;
  uint32_t stack_page = CallStackEntry::stack_page_from_base(ctx.base_memory_page);
  switch (mode) {
    case ZKW_MODE_REG:
    case ZKW_MODE_IMM:
      return false;  // :38-50
    case ZKW_MODE_STACK_PP:
      if (is_write) {  // :55-70 generalized push
        uint16_t old_sp = sp;
        sp = (uint16_t)(sp + vaddr);
        *loc = MemoryLocation{ZKW_MEM_STACK, stack_page, old_sp};
      } else {  // :71-86 generalized pop
        sp = (uint16_t)(sp - vaddr);
        *loc = MemoryLocation{ZKW_MEM_STACK, stack_page, sp};
      }
      return true;
    case ZKW_MODE_STACK_OFF:  // :88-98
      *loc = MemoryLocation{ZKW_MEM_STACK, stack_page, (uint16_t)(sp - vaddr)};
      return true;
    case ZKW_MODE_CODE:  // :100-110
      *loc = MemoryLocation{ZKW_MEM_CODE, ctx.code_page, vaddr};
      return true;
    case ZKW_MODE_STACK_ABS:  // :111-121
      *loc = MemoryLocation{ZKW_MEM_STACK, stack_page, vaddr};
      return true;
  }
  throw RefPanic("bad operand mode in ISA table");
}

static Decoded parse_opcode(const zkw_isa_table* isa, uint64_t enc, uint32_t* raw_idx) {
  // EncodingModeProduction::parse_preliminary_variant_and_absolute_number (absent crate;
  // bit layout per SURVEY Appendix B)
  Decoded d;
  uint32_t idx = (uint32_t)(enc & (ZKW_ISA_TABLE_SIZE - 1));
  *raw_idx = idx;
  d.variant = isa->entries[idx];
  // Condition::from the 3-bit field (absent crate): which Condition a field value names is a table constant — the truth
  // table of the field over (lt | eq << 1 | gt << 2) identifies it among the eight of cycle.rs:193-209
  {
    static const uint8_t truth[8] = {0xff /* Always */, 0xf0 /* Gt */, 0xaa /* Lt */, 0xcc /* Eq */, 0xfc /* Ge */, 0xee /* Le */, 0x33 /* Ne */, 0xfa /* GtOrLt */};
    const uint8_t t = (uint8_t)(isa->consts.condition_lut >> (8 * ((enc >> 13) & 7)));
    int which = -1;
    for (int c = 0; c < 8; c++) if (truth[c] == t) which = c;
    if (which < 0) throw RefPanic("condition_lut row names no Condition");
    d.condition = (uint8_t)which;
  }
  d.src0_reg_idx = (uint8_t)((enc >> 16) & 15);
  d.src1_reg_idx = (uint8_t)((enc >> 20) & 15);
  d.dst0_reg_idx = (uint8_t)((enc >> 24) & 15);
  d.dst1_reg_idx = (uint8_t)((enc >> 28) & 15);
  d.imm_0 = (uint16_t)(enc >> 32);
  d.imm_1 = (uint16_t)(enc >> 48);
  return d;
}
static void mask_into(const zkw_isa_table* isa, Decoded& d, uint32_t variant_idx) {
  // DecodedOpcode::mask_into_panic / mask_into_nop: "NOP r0, r0, r0, r0 after masking, or RET" (cycle.rs:295)
  d.variant = isa->entries[variant_idx];
  d.condition = 0;
  d.src0_reg_idx = d.src1_reg_idx = d.dst0_reg_idx = d.dst1_reg_idx = 0;
  d.imm_0 = d.imm_1 = 0;
}

enum { ERR_INVALID_OPCODE = 1, ERR_NOT_ENOUGH_ERGS = 2, ERR_PRIVILEGED = 4, ERR_WRITE_IN_STATIC = 8, ERR_CALLSTACK_FULL = 16 };  // helpers.rs:344-353

// cycle.rs:19-236; DelayedLocalStateChanges (mod.rs:110-153) are applied by the caller in
// the same order as `delayed_changes.apply` — here directly at the end, since nothing between
// reads the fields being changed.
Decoded Vm::read_and_decode(bool* skip_cycle_out) {
  witness_tracer.start_new_execution_cycle(local_state);  // :34
  bool execution_has_ended = local_state.execution_has_ended();
  bool pending_exception = local_state.pending_exception;
  const CallStackEntry& cs = local_state.callstack.current;
  uint32_t code_page = cs.code_page;
  uint32_t new_previous_code_memory_page = code_page;  // :49
  bool has_new_code_word = false, has_new_super_pc = false, has_new_pending = false;
  U256 new_code_word = U256::zero();
  uint16_t new_super_pc = 0;
  uint16_t pc = cs.pc;
  uint16_t previous_super_pc = local_state.previous_super_pc;
  bool code_pages_are_different = cs.code_page != local_state.previous_code_memory_page;
  uint16_t super_pc = (uint16_t)(pc >> 2);
  uint8_t sub_pc = (uint8_t)(pc & 3);  // E::split_pc :55 (cycle.rs:250-255)

  uint64_t opcode_encoding;
  auto sub_word = [](const U256& w, uint8_t sub) { return w.l[3 - sub]; };  // integer_representaiton_from_u256 (:86-94)
  if (!execution_has_ended && !pending_exception) {
    if (code_pages_are_different || previous_super_pc != super_pc) {  // :59-60
      MemoryLocation loc{ZKW_MEM_CODE, code_page, super_pc};
      MemoryQuery q = read_code(local_state.monotonic_cycle_counter, local_state.timestamp + 0, loc);  // :76-81
      new_code_word = q.value; has_new_code_word = true;  // :83
      new_super_pc = super_pc; has_new_super_pc = true;   // :84
      opcode_encoding = sub_word(q.value, sub_pc);
    } else {
      opcode_encoding = sub_word(local_state.previous_code_word, sub_pc);  // :96-100
    }
  } else if (pending_exception) {
    REF_ASSERT(execution_has_ended == false, "pending exception after end");  // :107
    has_new_pending = true;                                                   // :110
    new_super_pc = super_pc; has_new_super_pc = true;                         // :113
    opcode_encoding = isa->consts.exception_revert_encoding;                  // :115
  } else {
    opcode_encoding = isa->consts.nop_encoding;  // :126
  }
  bool skip_cycle = execution_has_ended;  // :129

  uint32_t error_flags = 0;
  uint32_t raw_idx;
  Decoded d = parse_opcode(isa, opcode_encoding, &raw_idx);  // :135-136
  if (d.variant.props & ZKW_PROP_EXPLICIT_PANIC) error_flags |= ERR_INVALID_OPCODE;  // :142-144
  uint32_t ergs_cost = isa->entries[raw_idx].price;  // :147-148
  if (skip_cycle) ergs_cost = 0;                     // :149-152
  uint32_t ergs_remaining;
  if (cs.ergs_remaining < ergs_cost) {  // :153-161
    ergs_remaining = 0;
    error_flags |= ERR_NOT_ENOUGH_ERGS;
  } else {
    ergs_remaining = cs.ergs_remaining - ergs_cost;
  }
  bool is_kernel = cs.is_kernel_mode();
  bool is_static_execution = cs.is_static;
  bool callstack_is_full = local_state.callstack.depth() == (size_t)isa->consts.vm_max_stack_depth;  // :172, execution_stack.rs:119-121
  if ((d.variant.props & ZKW_PROP_KERNEL_ONLY) && !is_kernel) error_flags |= ERR_PRIVILEGED;           // :174-176
  if (!(d.variant.props & ZKW_PROP_STATIC_OK) && is_static_execution) error_flags |= ERR_WRITE_IN_STATIC;  // :178-180
  if (callstack_is_full) error_flags |= ERR_CALLSTACK_FULL;                                           // :182-184
  bool mask_into_panic = error_flags != 0;  // :187
  if (mask_into_panic) mask_into(isa, d, isa->consts.panic_variant_idx);
  bool resolved;
  const Flags& f = local_state.flags;
  switch (d.condition) {  // :193-209, Condition order Always,Gt,Lt,Eq,Ge,Le,Ne,GtOrLt
    case 0: resolved = true; break;
    case 1: resolved = f.greater_than_flag; break;
    case 2: resolved = f.overflow_or_less_than_flag; break;
    case 3: resolved = f.equality_flag; break;
    case 4: resolved = f.greater_than_flag | f.equality_flag; break;
    case 5: resolved = f.overflow_or_less_than_flag | f.equality_flag; break;
    case 6: resolved = f.equality_flag == false; break;
    default: resolved = f.greater_than_flag | f.overflow_or_less_than_flag; break;
  }
  if (!resolved && !mask_into_panic) mask_into(isa, d, isa->consts.nop_variant_idx);  // :212-217

  // delayed_changes.apply(&mut self.local_state)  cycle.rs:267 / mod.rs:134-153
  local_state.callstack.current.ergs_remaining = ergs_remaining;
  if (has_new_code_word) local_state.previous_code_word = new_code_word;
  if (has_new_super_pc) local_state.previous_super_pc = new_super_pc;
  if (has_new_pending) local_state.pending_exception = false;
  local_state.previous_code_memory_page = new_previous_code_memory_page;
  *skip_cycle_out = skip_cycle;
  return d;
}

// zkevm_opcode_defs::utils::erase_fat_pointer_metadata (absent crate; Appendix B:
// clears page & start, keeps offset & length). Unreachable from kernel-mode tapes.
static void erase_fat_pointer_metadata(U256& v) {
  v.l[0] &= 0x00000000ffffffffULL;
  v.l[1] &= 0xffffffff00000000ULL;
}

// cycle.rs:257-429
void Vm::cycle() {
  bool skip_cycle;
  Decoded op = read_and_decode(&skip_cycle);  // :261-267

  uint16_t sp = local_state.callstack.current.sp;  // :275-277
  PrimitiveValue src0_reg_value, dummy;
  MemoryLocation src0_loc{0, 0, 0}, dst0_loc{0, 0, 0};
  bool has_src0_loc = compute_addresses(sp, op.src0_reg_idx, op.imm_0, op.variant.src0_mode, false, &src0_reg_value, &src0_loc);  // :278-285
  bool has_dst0_loc = compute_addresses(sp, op.dst0_reg_idx, op.imm_1, op.variant.dst0_mode, true, &dummy, &dst0_loc);            // :287-293
  local_state.callstack.current.sp = sp;                   // :297
  if (op.variant.opcode == ZKW_OP_NOP) has_src0_loc = false;  // :298-301

  PrimitiveValue src0_mem_value = PrimitiveValue::empty();
  if (has_src0_loc) {  // :304-325
    MemoryQuery q = src0_loc.memory_type == ZKW_MEM_CODE ? read_code(local_state.monotonic_cycle_counter, timestamp_for_code_or_src_read(), src0_loc)
                                                         : read_memory(local_state.monotonic_cycle_counter, timestamp_for_code_or_src_read(), src0_loc);
    src0_mem_value = PrimitiveValue{q.value, q.value_is_pointer};
  }
  PrimitiveValue src0;
  switch (op.variant.src0_mode) {  // :327-337
    case ZKW_MODE_REG: src0 = src0_reg_value; break;
    case ZKW_MODE_IMM: src0 = PrimitiveValue{U256::from_u64(op.imm_0), false}; break;
    default: src0 = src0_mem_value; break;
  }
  PrimitiveValue src1 = select_register_value(op.src1_reg_idx);  // :339
  if (op.variant.props & ZKW_PROP_SWAP) std::swap(src0, src1);   // :341-345
  uint16_t new_pc = local_state.callstack.current.pc;
  if (!skip_cycle) new_pc = (uint16_t)(new_pc + 1);  // :347-350
  bool is_kernel_mode = local_state.callstack.current.is_kernel_mode();  // :368-372
  if (!(op.variant.props & ZKW_PROP_SRC0_PTR_OK) && src0.is_pointer && !is_kernel_mode) {  // :375-385
    erase_fat_pointer_metadata(src0.value);
    src0.is_pointer = false;
  }
  if (!(op.variant.props & ZKW_PROP_SRC1_PTR_OK) && src1.is_pointer && !is_kernel_mode) {  // :386-396
    erase_fat_pointer_metadata(src1.value);
    src1.is_pointer = false;
  }
  PreState ps{src0, src1, has_dst0_loc, dst0_loc, new_pc, is_kernel_mode};  // :398-404
  apply(op, ps);                                                             // :406
  if (!skip_cycle) local_state.timestamp += isa->consts.time_delta_per_cycle;  // :408-410
  local_state.monotonic_cycle_counter += 1;                                    // :411
  witness_tracer.end_execution_cycle(local_state);                            // :413
}

// opcodes/parsing.rs:47-79
void Vm::apply(const Decoded& op, const PreState& ps) {
  switch (op.variant.opcode) {
    case ZKW_OP_NOP: local_state.callstack.current.pc = ps.new_pc; break;  // noop.rs:16-19
    case ZKW_OP_ADD: add_sub(op, ps, false); break;
    case ZKW_OP_SUB: add_sub(op, ps, true); break;
    case ZKW_OP_MUL: mul(op, ps); break;
    case ZKW_OP_DIV: div(op, ps); break;
    case ZKW_OP_JUMP: local_state.callstack.current.pc = clip16(ps.src0.value.low_u64()); break;  // jump.rs:23-25
    case ZKW_OP_CONTEXT: context(op, ps); break;
    case ZKW_OP_SHIFT: shift(op, ps); break;
    case ZKW_OP_BINOP: binop(op, ps); break;
    case ZKW_OP_PTR: ptr(op, ps); break;
    case ZKW_OP_LOG: log(op, ps); break;
    case ZKW_OP_NEAR_CALL: near_call(op, ps); break;
    case ZKW_OP_FAR_CALL: far_call(op, ps); break;
    case ZKW_OP_RET: ret(op, ps); break;
    case ZKW_OP_UMA: uma(op, ps); break;
    default: throw RefPanic("unreachable: Opcode::Invalid in apply");  // parsing.rs:77
  }
}

// add.rs:4-54, sub.rs:4-55
void Vm::add_sub(const Decoded& op, const PreState& ps, bool is_sub) {
  bool set_flags = op.variant.flags & 1;
  local_state.callstack.current.pc = ps.new_pc;
  bool of;
  U256 result = is_sub ? overflowing_sub(ps.src0.value, ps.src1.value, of) : overflowing_add(ps.src0.value, ps.src1.value, of);
  bool eq = result.is_zero();
  bool gt = !eq && !of;
  if (set_flags) {
    local_state.flags.overflow_or_less_than_flag = of;
    local_state.flags.equality_flag = eq;
    local_state.flags.greater_than_flag = gt;
  }
  perform_dst0_update(local_state.monotonic_cycle_counter, PrimitiveValue{result, false}, ps, op);
}

// mul.rs:4-66
void Vm::mul(const Decoded& op, const PreState& ps) {
  bool set_flags = op.variant.flags & 1;
  local_state.callstack.current.pc = ps.new_pc;
  uint64_t tmp[8];
  full_mul(ps.src0.value, ps.src1.value, tmp);
  U256 low{{tmp[0], tmp[1], tmp[2], tmp[3]}}, high{{tmp[4], tmp[5], tmp[6], tmp[7]}};
  if (set_flags) {
    bool of = !high.is_zero();
    bool eq = low.is_zero();
    local_state.flags.reset();
    local_state.flags.overflow_or_less_than_flag = of;
    local_state.flags.equality_flag = eq;
    local_state.flags.greater_than_flag = !of & !eq;
  }
  perform_dst0_update(local_state.monotonic_cycle_counter, PrimitiveValue{low, false}, ps, op);
  perform_dst1_update(PrimitiveValue{high, false}, op.dst1_reg_idx);
}

// div.rs:4-76
void Vm::div(const Decoded& op, const PreState& ps) {
  bool set_flags = op.variant.flags & 1;
  local_state.callstack.current.pc = ps.new_pc;
  if (ps.src1.value.is_zero()) {
    if (set_flags) {
      local_state.flags.reset();
      local_state.flags.overflow_or_less_than_flag = true;
    }
    perform_dst0_update(local_state.monotonic_cycle_counter, PrimitiveValue::empty(), ps, op);
    perform_dst1_update(PrimitiveValue::empty(), op.dst1_reg_idx);
  } else {
    U256 q, r;
    div_mod(ps.src0.value, ps.src1.value, q, r);
    if (set_flags) {
      bool eq = q.is_zero();
      bool gt = r.is_zero();
      local_state.flags.reset();
      local_state.flags.equality_flag = eq;
      local_state.flags.greater_than_flag = gt;
    }
    perform_dst0_update(local_state.monotonic_cycle_counter, PrimitiveValue{q, false}, ps, op);
    perform_dst1_update(PrimitiveValue{r, false}, op.dst1_reg_idx);
  }
}

// shift.rs:8-79
void Vm::shift(const Decoded& op, const PreState& ps) {
  bool set_flags = op.variant.flags & 1;
  local_state.callstack.current.pc = ps.new_pc;
  uint32_t shift_abs = (uint8_t)ps.src1.value.low_u64();  // :44
  uint8_t v = op.variant.variant;
  bool is_cyclic = v == ZKW_SHIFT_ROL || v == ZKW_SHIFT_ROR;
  bool is_right_shift = v == ZKW_SHIFT_SHR || v == ZKW_SHIFT_ROR;
  U256 result;
  if (is_right_shift) {
    result = shr(ps.src0.value, shift_abs);
    if (is_cyclic) result = bit_or(result, shl(ps.src0.value, 256u - shift_abs));
  } else {
    result = shl(ps.src0.value, shift_abs);
    if (is_cyclic) result = bit_or(result, shr(ps.src0.value, 256u - shift_abs));
  }
  if (set_flags) {
    bool eq = result.is_zero();
    local_state.flags.reset();
    local_state.flags.equality_flag = eq;
  }
  perform_dst0_update(local_state.monotonic_cycle_counter, PrimitiveValue{result, false}, ps, op);
}

// binop.rs:5-62
void Vm::binop(const Decoded& op, const PreState& ps) {
  bool set_flags = op.variant.flags & 1;
  local_state.callstack.current.pc = ps.new_pc;
  U256 result;
  switch (op.variant.variant) {
    case ZKW_BINOP_XOR: result = bit_xor(ps.src0.value, ps.src1.value); break;
    case ZKW_BINOP_AND: result = bit_and(ps.src0.value, ps.src1.value); break;
    default: result = bit_or(ps.src0.value, ps.src1.value); break;
  }
  if (set_flags) {
    bool eq = result.is_zero();
    local_state.flags.reset();
    local_state.flags.equality_flag = eq;
  }
  perform_dst0_update(local_state.monotonic_cycle_counter, PrimitiveValue{result, false}, ps, op);
}

// ptr.rs:6-194
void Vm::ptr(const Decoded& op, const PreState& ps) {
  local_state.callstack.current.pc = ps.new_pc;
  uint8_t v = op.variant.variant;
  if (ps.src0.is_pointer == false) { set_shorthand_panic(); return; }  // :35-39,:98-102,:141-145
  if (ps.src1.is_pointer == true) { set_shorthand_panic(); return; }   // :41-45,:104-108,:147-151
  const U256& src0 = ps.src0.value;
  const U256& src1 = ps.src1.value;
  U256 result;
  if (v == ZKW_PTR_ADD || v == ZKW_PTR_SUB) {
    // ptr::MAX_OFFSET_FOR_ADD_SUB (2^32, Appendix B): a table constant
    if (src1.l[0] >= isa->consts.max_offset_for_add_sub || src1.l[1] || src1.l[2] || src1.l[3]) { set_shorthand_panic(); return; }  // :47-51
    FatPointer fp = FatPointer::from_u256(src0);
    uint32_t offset = src1.low_u32();
    uint32_t new_off;
    bool error;
    if (v == ZKW_PTR_ADD) { new_off = fp.offset + offset; error = new_off < fp.offset; }
    else { new_off = fp.offset - offset; error = fp.offset < offset; }
    if (error) { set_shorthand_panic(); return; }  // :71-74
    fp.offset = new_off;
    U256 p = fp.to_u256();
    result = U256{{p.l[0], p.l[1], src0.l[2], src0.l[3]}};  // :82
  } else if (v == ZKW_PTR_PACK) {
    if (src1.l[0] != 0 || src1.l[1] != 0) { set_shorthand_panic(); return; }  // :110-114
    result = U256{{src0.l[0], src0.l[1], src1.l[2], src1.l[3]}};               // :126
  } else {  // Shrink :139-191
    FatPointer fp = FatPointer::from_u256(src0);
    uint32_t offset = src1.low_u32();
    if (fp.length < offset) { set_shorthand_panic(); return; }  // :165-170
    fp.length -= offset;
    U256 p = fp.to_u256();
    result = U256{{p.l[0], p.l[1], src0.l[2], src0.l[3]}};
  }
  perform_dst0_update(local_state.monotonic_cycle_counter, PrimitiveValue{result, true}, ps, op);
}

// context.rs:6-111
void Vm::context(const Decoded& op, const PreState& ps) {
  local_state.callstack.current.pc = ps.new_pc;
  const CallStackEntry& c = local_state.callstack.current;
  const U256& src0 = ps.src0.value;
  U256 value;
  switch (op.variant.variant) {
    case ZKW_CTX_SET_CONTEXT_U128:
      local_state.context_u128_register[0] = src0.l[0];
      local_state.context_u128_register[1] = src0.l[1];
      return;
    case ZKW_CTX_SET_ERGS_PER_PUBDATA: local_state.current_ergs_per_pubdata_byte = src0.low_u32(); return;
    case ZKW_CTX_INC_TX_NUMBER: local_state.tx_number_in_block = (uint16_t)(local_state.tx_number_in_block + 1); return;
    case ZKW_CTX_THIS: value = address_to_u256(c.this_address); break;
    case ZKW_CTX_CALLER: value = address_to_u256(c.msg_sender); break;
    case ZKW_CTX_CODE_ADDRESS: value = address_to_u256(c.code_address); break;
    case ZKW_CTX_META: {
      // VmMetaParameters::to_u256 (absent crate; Appendix B layout)
      value = U256::zero();
      value.l[0] = local_state.current_ergs_per_pubdata_byte;
      value.l[1] = (uint64_t)c.heap_bound | ((uint64_t)c.aux_heap_bound << 32);
      value.l[3] = ((uint64_t)c.this_shard_id << 32) | ((uint64_t)c.caller_shard_id << 40) | ((uint64_t)c.code_shard_id << 48);
      break;
    }
    case ZKW_CTX_ERGS_LEFT: value = U256::from_u64(c.ergs_remaining); break;
    case ZKW_CTX_SP: value = U256::from_u64(c.sp); break;
    case ZKW_CTX_GET_CONTEXT_U128: value = U256::from_u128(c.context_u128_value[0], c.context_u128_value[1]); break;
    default: throw RefPanic("bad context variant");
  }
  perform_dst0_update(local_state.monotonic_cycle_counter, PrimitiveValue{value, false}, ps, op);
}

// near_call.rs:6-68
void Vm::near_call(const Decoded& op, const PreState& ps) {
  local_state.flags.reset();
  uint16_t dst = op.imm_0;
  uint16_t exception_handler_location = op.imm_1;
  uint32_t abi_ergs_passed = ps.src0.value.low_u32();  // NearCallABI::from_u256
  bool pass_all_ergs = abi_ergs_passed == 0;
  uint32_t remaining_ergs = local_state.callstack.current.ergs_remaining;
  uint32_t passed_ergs, remaining_for_this;
  if (pass_all_ergs) {
    passed_ergs = remaining_ergs; remaining_for_this = 0;
  } else if (remaining_ergs < abi_ergs_passed) {
    passed_ergs = remaining_ergs; remaining_for_this = 0;
  } else {
    passed_ergs = abi_ergs_passed; remaining_for_this = remaining_ergs - abi_ergs_passed;
  }
  local_state.callstack.current.ergs_remaining = remaining_for_this;
  local_state.callstack.current.pc = ps.new_pc;
  CallStackEntry new_stack = local_state.callstack.current;
  new_stack.pc = dst;
  new_stack.exception_handler_location = exception_handler_location;
  new_stack.ergs_remaining = passed_ergs;
  new_stack.is_local_frame = true;
  start_frame(local_state.monotonic_cycle_counter, new_stack);
}

// helpers.rs:196-223 + DefaultPrecompilesProcessor::execute_precompile (absent crate; dispatch on
// the low 16 bits of the address, Appendix B)
void Vm::call_precompile(uint32_t cc, const LogQuery& query) {
  witness_tracer.add_log(query, ZKW_LQ_LOG, cc);
  uint32_t address_low = (uint32_t)query.address.b[0] | ((uint32_t)query.address.b[1] << 8);
  std::vector<MemoryQuery> reads, writes;
  // round witness: (reads consumed, writes performed) per round; round 0 carries the request (PrecompileCyclesWitness::
  // {Sha256, Keccak256, ECRecover}(Vec<..RoundWitness{new_request, reads, writes}>), crate zk_evm_abstractions — recalled)
  std::vector<std::pair<uint32_t, uint32_t>> rounds;
  uint32_t kind;
  if (address_low == isa->consts.keccak_precompile_address)
    kind = 1, keccak256_rounds_function(cc, query, memory, reads, writes, rounds);
  else if (address_low == isa->consts.sha256_precompile_address)
    kind = 0, sha256_rounds_function(cc, query, memory, reads, writes, rounds);
  else if (address_low == isa->consts.ecrecover_precompile_address)
    kind = 2, ecrecover_function(cc, query, memory, isa->consts.ecrecover_input_layout, reads, writes, rounds);
  else
    return;  // unknown precompile address: the default processor does nothing
  if (witness_tracer.cb) {  // add_precompile_call_result(cc, query, mem_in, mem_out, round_witness) helpers.rs:214-221
    std::vector<cblog::MemQ> in, out;
    for (const MemoryQuery& q : reads) in.push_back(Recorder::cb_mem(q));
    for (const MemoryQuery& q : writes) out.push_back(Recorder::cb_mem(q));
    std::vector<cblog::Log::Round> rw;
    for (size_t r = 0; r < rounds.size(); r++) rw.push_back(cblog::Log::Round{(uint8_t)(r == 0), rounds[r].first, rounds[r].second});
    witness_tracer.cb->precompile(cc, Recorder::cb_log(query), in, out, kind, rw);
  }
  for (const MemoryQuery& q : reads) witness_tracer.add_memory_query(q, 1);
  for (const MemoryQuery& q : writes) witness_tracer.add_memory_query(q, 2);
}

// log.rs:11-330
void Vm::log(const Decoded& op, const PreState& ps) {
  const U256& src0 = ps.src0.value;
  const U256& src1 = ps.src1.value;
  uint8_t v = op.variant.variant;
  local_state.callstack.current.pc = ps.new_pc;
  bool is_first_message = op.variant.flags & 1;
  uint8_t shard_id = local_state.callstack.current.this_shard_id;
  uint32_t ergs_available = local_state.callstack.current.ergs_remaining;
  bool is_rollup = shard_id == 0;
  uint32_t timestamp_for_log = timestamp_for_first_decommit_or_precompile_read();
  uint16_t tx_number_in_block = local_state.tx_number_in_block;
  const zkw_isa_consts& K = isa->consts;
  uint32_t cc = local_state.monotonic_cycle_counter;

  uint32_t ergs_on_pubdata = 0;
  if (v == ZKW_LOG_STORAGE_WRITE) {  // :71-118
    LogQuery pq{timestamp_for_log, tx_number_in_block, K.storage_aux_byte, local_state.callstack.current.this_shard_id,
                local_state.callstack.current.this_address, src0, U256::zero(), src1, true, false, false};
    uint32_t pubdata_refund = refund_for_partial_query(cc, pq);
    uint32_t net_pubdata;
    if (is_rollup) {
      REF_ASSERT(K.initial_storage_write_pubdata_bytes >= pubdata_refund, "refund can not be more than net cost itself");
      net_pubdata = K.initial_storage_write_pubdata_bytes - pubdata_refund;
    } else {
      REF_ASSERT(pubdata_refund == 0, "porter refund");
      net_pubdata = 0;
    }
    ergs_on_pubdata = local_state.current_ergs_per_pubdata_byte * net_pubdata;
  } else if (v == ZKW_LOG_TO_L1) {
    ergs_on_pubdata = local_state.current_ergs_per_pubdata_byte * K.l1_message_pubdata_bytes;  // :120-124
  }
  uint32_t extra_cost = v == ZKW_LOG_PRECOMPILE ? src1.low_u32() : 0;  // :128-131
  uint32_t total_cost = extra_cost + ergs_on_pubdata;                  // :133 (u32 wrapping in release)
  bool not_enough_power = ergs_available < total_cost;
  uint32_t ergs_remaining = ergs_available - total_cost;
  if (not_enough_power) {  // :136-153
    local_state.callstack.current.ergs_remaining = 0;
    local_state.spent_pubdata_counter += ergs_available < ergs_on_pubdata ? ergs_available : ergs_on_pubdata;
  } else {
    local_state.callstack.current.ergs_remaining = ergs_remaining;
    local_state.spent_pubdata_counter += ergs_on_pubdata;
  }
  Address address = local_state.callstack.current.this_address;
  shard_id = local_state.callstack.current.this_shard_id;

  switch (v) {
    case ZKW_LOG_STORAGE_READ: {  // :163-195
      REF_ASSERT(not_enough_power == false, "storage read out of ergs");
      LogQuery pq{timestamp_for_log, tx_number_in_block, K.storage_aux_byte, shard_id, address, src0, U256::zero(), U256::zero(), false, false, is_first_message};
      LogQuery q = access_storage(cc, pq);
      perform_dst0_update(cc, PrimitiveValue{q.read_value, false}, ps, op);
      break;
    }
    case ZKW_LOG_STORAGE_WRITE: {  // :196-220
      if (not_enough_power) return;
      LogQuery pq{timestamp_for_log, tx_number_in_block, K.storage_aux_byte, shard_id, address, src0, U256::zero(), src1, true, false, is_first_message};
      access_storage(cc, pq);
      break;
    }
    case ZKW_LOG_EVENT:
    case ZKW_LOG_TO_L1: {  // :221-251
      if (not_enough_power) {
        REF_ASSERT(v == ZKW_LOG_TO_L1, "event out of ergs");
        return;
      }
      uint8_t aux_byte = v == ZKW_LOG_EVENT ? K.event_aux_byte : K.l1_message_aux_byte;
      LogQuery q{timestamp_for_log, tx_number_in_block, aux_byte, shard_id, address, src0, U256::zero(), src1, true, false, is_first_message};
      emit_event(cc, q);
      break;
    }
    case ZKW_LOG_PRECOMPILE: {  // :252-328
      if (not_enough_power) {
        perform_dst0_update(cc, PrimitiveValue::empty(), ps, op);
        return;
      }
      // PrecompileCallABI::from_u256 / to_u256 (absent crate; Appendix B layout): only the two
      // page fields (bits 128-159 / 160-191) are rewritten
      U256 abi = src0;
      local_state.callstack.current.ergs_remaining = ergs_remaining;
      uint32_t heap_page = CallStackEntry::heap_page_from_base(local_state.callstack.current.base_memory_page);
      if ((uint32_t)abi.l[2] == 0) abi.l[2] = (abi.l[2] & 0xffffffff00000000ULL) | heap_page;                     // :273-283
      if ((uint32_t)(abi.l[2] >> 32) == 0) abi.l[2] = (abi.l[2] & 0x00000000ffffffffULL) | ((uint64_t)heap_page << 32);  // :285-295
      LogQuery q{timestamp_for_log, tx_number_in_block, K.precompile_aux_byte, shard_id, address, abi, U256::zero(), U256::zero(), false, false, is_first_message};
      call_precompile(cc, q);
      perform_dst0_update(cc, PrimitiveValue{U256::from_u64(1), false}, ps, op);
      break;
    }
    default: throw RefPanic("bad log variant");
  }
}

enum {
  FC_INPUT_IS_NOT_POINTER_WHEN_EXPECTED = 1, FC_INVALID_CODE_HASH_FORMAT = 2, FC_NOT_ENOUGH_ERGS_TO_DECOMMIT = 4,
  FC_NOT_ENOUGH_ERGS_TO_GROW_MEMORY = 8, FC_MALFORMED_ABI_QUASI_POINTER = 16, FC_CALL_IN_NOW_CONSTRUCTED_SYSTEM_CONTRACT = 32,
  FC_NOT_ENOUGH_ERGS_FOR_EXTRA_FAR_CALL_COSTS = 64
};  // far_call.rs:15-25
enum { FWD_USE_HEAP = 0, FWD_FORWARD_FAT_POINTER = 1, FWD_USE_AUX_HEAP = 2 };  // FarCallForwardPageType / RetForwardPageType
// FarCallForwardPageType::from_u8 (absent crate): the byte codes are table constants (forwarding_codes), anything else is UseHeap
static int forward_type_from_u8(uint32_t codes, uint8_t b) { return b == ((codes >> 8) & 0xff) ? FWD_FORWARD_FAT_POINTER : (b == ((codes >> 16) & 0xff) ? FWD_USE_AUX_HEAP : FWD_USE_HEAP); }

// VersionedHashGeneric<ContractCodeSha256> (absent crate; Appendix B): byte0 = version 1,
// byte1 = marker (0 at rest, 1 yet constructed), bytes 2-3 = length in words BE
struct VersionedHash {
  bool ok;
  uint8_t marker;
  uint16_t code_length_in_words;
  U256 stored;  // serialize_to_stored: marker forced to CODE_AT_REST
};
static VersionedHash versioned_hash_from(const U256& h) {
  VersionedHash r;
  uint8_t buf[32];
  to_big_endian(h, buf);
  r.ok = buf[0] == 1;
  r.marker = buf[1];
  r.code_length_in_words = (uint16_t)((buf[2] << 8) | buf[3]);
  buf[1] = 0;
  r.stored = from_big_endian(buf);
  return r;
}

// far_call.rs:35-613
void Vm::far_call(const Decoded& op, const PreState& ps) {
  const zkw_isa_consts& K = isa->consts;
  uint8_t inner_variant = op.variant.variant;
  U256 abi_src = ps.src0.value;
  bool abi_src_is_ptr = ps.src0.is_pointer;
  U256 call_destination_value = ps.src1.value;
  local_state.flags.reset();  // :69
  bool is_static_call = op.variant.flags & 1;  // FAR_CALL_STATIC_FLAG_IDX
  bool is_call_shard = op.variant.flags & 2;   // FAR_CALL_SHARD_FLAG_IDX
  uint16_t exception_handler_location = op.imm_0;
  Address called_address = u256_to_address_unchecked(call_destination_value);
  U256 called_address_as_u256 = address_to_u256(called_address);  // & U256_TO_ADDRESS_MASK (:78)
  bool dst_is_kernel = address_is_kernel(called_address);

  // FarCallABI::from_u256 (absent crate; Appendix B layout)
  FatPointer abi_ptr = FatPointer::from_u256(abi_src);
  uint32_t abi_ergs_passed = (uint32_t)abi_src.l[3];
  int forwarding_mode = forward_type_from_u8(K.forwarding_codes, (uint8_t)(abi_src.l[3] >> 32));
  uint8_t abi_shard_id = (uint8_t)(abi_src.l[3] >> 40);
  bool constructor_call = ((uint8_t)(abi_src.l[3] >> 48)) != 0;
  bool to_system = ((uint8_t)(abi_src.l[3] >> 56)) != 0;
  constructor_call = constructor_call & ps.is_kernel_mode;  // :85
  to_system = to_system & dst_is_kernel;                    // :86

  const CallStackEntry cs = local_state.callstack.current;
  Address current_address = cs.this_address, current_msg_sender = cs.msg_sender;
  uint32_t current_base_page = cs.base_memory_page;
  uint8_t caller_shard_id = cs.this_shard_id;
  uint32_t remaining_ergs = cs.ergs_remaining;
  uint64_t current_context_u128[2] = {cs.context_u128_value[0], cs.context_u128_value[1]};
  uint32_t timestamp_for_storage_read = timestamp_for_first_decommit_or_precompile_read();
  uint16_t tx_number_in_block = local_state.tx_number_in_block;
  uint8_t new_code_shard_id = is_call_shard ? abi_shard_id : caller_shard_id;                             // :105-109
  uint8_t new_this_shard_id = inner_variant == ZKW_FAR_DELEGATE ? caller_shard_id : new_code_shard_id;  // :112-116
  uint32_t new_base_memory_page = local_state.memory_page_counter;                                        // :118
  uint32_t cc = local_state.monotonic_cycle_counter;

  U256 code_hash;
  bool map_to_trivial;
  if (new_code_shard_id != 0 && !block_properties.zkporter_is_available) {  // :123-129
    code_hash = U256::zero();
    map_to_trivial = true;
  } else {
    LogQuery pq{timestamp_for_storage_read, tx_number_in_block, K.storage_aux_byte, new_code_shard_id, address_from_low_u32(K.deployer_address_low),
                called_address_as_u256, U256::zero(), U256::zero(), false, false, false};
    LogQuery q = access_storage(cc, pq);  // :144-145
    U256 code_hash_from_storage = q.read_value;
    bool mask_into_default_aa = code_hash_from_storage.is_zero() && dst_is_kernel == false;
    code_hash = mask_into_default_aa ? block_properties.default_aa_code_hash : code_hash_from_storage;
    map_to_trivial = false;
  }
  uint32_t memory_page_candidate_for_code_decommittment = map_to_trivial ? 0u : CallStackEntry::code_page_candidate_from_base(new_base_memory_page);  // :161-165

  uint32_t exceptions = 0;
  uint32_t code_length_in_words = 0;
  VersionedHash vh = versioned_hash_from(code_hash);  // :171-178
  if (vh.ok) {
    bool at_rest = vh.marker == 0, constructed_now = vh.marker == 1;
    if (!(at_rest || constructed_now)) {  // :192-196
      exceptions |= FC_INVALID_CODE_HASH_FORMAT;
      code_hash = U256::zero();
      code_length_in_words = 0;
    } else {
      bool can_call_at_rest = !constructor_call && at_rest;
      bool can_call_by_constructor = constructor_call && constructed_now;
      if (can_call_at_rest || can_call_by_constructor) {  // :209-211
        code_hash = vh.stored;
        code_length_in_words = vh.code_length_in_words;
      } else if (dst_is_kernel == false) {  // :215-237
        VersionedHash aa = versioned_hash_from(block_properties.default_aa_code_hash);
        REF_ASSERT(aa.ok, "default AA code hash must be always valid");
        REF_ASSERT(aa.marker == 0, "default AA marker is always in storage format");
        code_hash = block_properties.default_aa_code_hash;
        code_length_in_words = aa.code_length_in_words;
      } else {  // :238-245
        exceptions |= FC_CALL_IN_NOW_CONSTRUCTED_SYSTEM_CONTRACT;
        code_hash = U256::zero();
        code_length_in_words = 0;
      }
    }
  } else {  // :248-252
    exceptions |= FC_INVALID_CODE_HASH_FORMAT;
    code_hash = U256::zero();
    code_length_in_words = 0;
  }
  if (forwarding_mode == FWD_FORWARD_FAT_POINTER && abi_src_is_ptr == false) exceptions |= FC_INPUT_IS_NOT_POINTER_WHEN_EXPECTED;  // :255-262
  bool validate_as_fresh = forwarding_mode != FWD_FORWARD_FAT_POINTER;
  uint32_t pointer_validation_exceptions = fat_pointer_validate(abi_ptr, validate_as_fresh);  // :271-273
  if (pointer_validation_exceptions != 0) exceptions |= FC_MALFORMED_ABI_QUASI_POINTER;
  if (abi_ptr.validate_as_slice() == false) exceptions |= FC_MALFORMED_ABI_QUASI_POINTER;  // :280-282
  switch (forwarding_mode) {  // :285-314
    case FWD_FORWARD_FAT_POINTER: {
      uint32_t new_start = abi_ptr.start + abi_ptr.offset;
      uint32_t new_length = abi_ptr.length - abi_ptr.offset;
      abi_ptr.start = new_start; abi_ptr.length = new_length; abi_ptr.offset = 0;
      break;
    }
    case FWD_USE_HEAP: abi_ptr.memory_page = CallStackEntry::heap_page_from_base(current_base_page); break;
    default: abi_ptr.memory_page = CallStackEntry::aux_heap_page_from_base(current_base_page); break;
  }
  if (exceptions != 0) abi_ptr = FatPointer::empty();  // :321-325

  uint32_t memory_growth_in_bytes = 0;  // :330-369
  if (forwarding_mode == FWD_USE_HEAP || forwarding_mode == FWD_USE_AUX_HEAP) {
    uint32_t upper_bound = abi_ptr.start + abi_ptr.length;  // validated pointer cannot overflow (:332-334); malformed ones were zeroed above
    if (pointer_validation_exceptions & FPV_DEREF_BEYOND_HEAP_RANGE) upper_bound = 0xffffffffu;
    CallStackEntry& m = local_state.callstack.current;
    uint32_t current_bound = forwarding_mode == FWD_USE_HEAP ? m.heap_bound : m.aux_heap_bound;
    if (upper_bound < current_bound) {
      memory_growth_in_bytes = 0;
    } else {
      memory_growth_in_bytes = upper_bound - current_bound;
      if (forwarding_mode == FWD_USE_HEAP) m.heap_bound = upper_bound; else m.aux_heap_bound = upper_bound;
    }
  }
  uint32_t cost_of_memory_growth = memory_growth_in_bytes * K.memory_growth_ergs_per_byte;  // :375-376 wrapping_mul
  uint32_t remaining_ergs_after_growth;
  if (remaining_ergs >= cost_of_memory_growth) remaining_ergs_after_growth = remaining_ergs - cost_of_memory_growth;
  else { exceptions |= FC_NOT_ENOUGH_ERGS_TO_GROW_MEMORY; remaining_ergs_after_growth = 0; }
  uint32_t msg_value_stipend = 0;  // FORCED_ERGS_FOR_MSG_VALUE_SIMULATOR == false (:13, :387-406)
  uint32_t remaining_ergs_of_caller_frame = remaining_ergs_after_growth - msg_value_stipend;  // :408-420
  uint32_t cost_of_decommittment = K.ergs_per_code_word_decommittment * code_length_in_words;  // :423-424
  uint32_t remaining_ergs_after_decommittment;
  if (remaining_ergs_of_caller_frame >= cost_of_decommittment) remaining_ergs_after_decommittment = remaining_ergs_of_caller_frame - cost_of_decommittment;
  else { exceptions |= FC_NOT_ENOUGH_ERGS_TO_DECOMMIT; remaining_ergs_after_decommittment = remaining_ergs_of_caller_frame; }
  uint32_t mapped_code_page;
  if (exceptions != 0) {  // :435-439
    set_shorthand_panic();
    mapped_code_page = K.unmapped_page;  // UNMAPPED_PAGE
  } else {  // :441-455
    DecommittmentQuery dq = decommit(cc, code_hash, memory_page_candidate_for_code_decommittment, timestamp_for_first_decommit_or_precompile_read());
    if (dq.is_fresh == false) remaining_ergs_after_decommittment += cost_of_decommittment;
    mapped_code_page = dq.memory_page;
  }
  uint32_t stipend_for_callee = msg_value_stipend;

  uint32_t remaining_ergs_to_pass = remaining_ergs_after_decommittment;  // :468-484
  uint32_t max_passable = (remaining_ergs_to_pass / 64) * 63;
  uint32_t leftover = remaining_ergs_to_pass - max_passable;
  uint32_t passed_ergs, remaining_ergs_for_this_context;
  if (max_passable < abi_ergs_passed) { passed_ergs = max_passable; remaining_ergs_for_this_context = leftover; }
  else { passed_ergs = abi_ergs_passed; remaining_ergs_for_this_context = leftover + (max_passable - abi_ergs_passed); }
  passed_ergs = passed_ergs + stipend_for_callee;  // :487

  local_state.callstack.current.ergs_remaining = remaining_ergs_for_this_context;  // :490-495
  local_state.callstack.current.pc = ps.new_pc;
  bool new_context_is_static = local_state.callstack.current.is_static | is_static_call;  // :500
  local_state.memory_page_counter += K.new_memory_pages_per_far_call;                     // :503
  const uint32_t CALL_IMPLICIT_PARAMETER_REG_IDX = (K.call_regs >> 16) & 0xff;  // (table constants: indices into `registers`)
  Address address_from_implicit_reg = u256_to_address_unchecked(local_state.registers[CALL_IMPLICIT_PARAMETER_REG_IDX].value);  // :506-508
  Address address_for_next, msg_sender_for_next;
  switch (inner_variant) {  // :510-523
    case ZKW_FAR_NORMAL: address_for_next = called_address; msg_sender_for_next = current_address; break;
    case ZKW_FAR_DELEGATE: address_for_next = current_address; msg_sender_for_next = current_msg_sender; break;
    default: address_for_next = called_address; msg_sender_for_next = address_from_implicit_reg; break;
  }
  uint64_t context_u128_for_next[2];
  if (inner_variant == ZKW_FAR_DELEGATE) { context_u128_for_next[0] = current_context_u128[0]; context_u128_for_next[1] = current_context_u128[1]; }
  else { context_u128_for_next[0] = local_state.context_u128_register[0]; context_u128_for_next[1] = local_state.context_u128_register[1]; }
  CallStackEntry new_stack;  // :537-555
  new_stack.this_address = address_for_next; new_stack.msg_sender = msg_sender_for_next; new_stack.code_address = called_address;
  new_stack.base_memory_page = new_base_memory_page; new_stack.code_page = mapped_code_page;
  new_stack.sp = clip16(K.initial_sp_on_far_call); new_stack.pc = 0; new_stack.exception_handler_location = exception_handler_location;
  new_stack.ergs_remaining = passed_ergs;
  new_stack.this_shard_id = new_this_shard_id; new_stack.caller_shard_id = caller_shard_id; new_stack.code_shard_id = new_code_shard_id;
  new_stack.is_static = new_context_is_static; new_stack.is_local_frame = false;
  new_stack.context_u128_value[0] = context_u128_for_next[0]; new_stack.context_u128_value[1] = context_u128_for_next[1];
  new_stack.heap_bound = K.new_frame_memory_stipend; new_stack.aux_heap_bound = K.new_frame_memory_stipend;
  local_state.context_u128_register[0] = local_state.context_u128_register[1] = 0;  // :558
  start_frame(cc, new_stack);                                                        // :562
  memory.start_global_frame(current_base_page, new_base_memory_page, abi_ptr, local_state.timestamp);  // :564-569

  const uint32_t CALL_IMPLICIT_CALLDATA_FAT_PTR_REGISTER = K.call_regs & 0xff, CALL_IMPLICIT_CONSTRUCTOR_MARKER_REGISTER = (K.call_regs >> 8) & 0xff;
  const uint32_t abi_first = K.call_ranges & 0xff, abi_end = (K.call_ranges >> 8) & 0xff, res_first = (K.call_ranges >> 16) & 0xff, res_end = K.call_ranges >> 24;
  local_state.registers[CALL_IMPLICIT_CALLDATA_FAT_PTR_REGISTER] = PrimitiveValue{abi_ptr.to_u256(), true};  // :573-577
  U256 r2 = U256::zero();
  if (constructor_call) r2.l[0] |= 1;
  if (to_system) r2.l[0] |= 2;
  local_state.registers[CALL_IMPLICIT_CONSTRUCTOR_MARKER_REGISTER] = PrimitiveValue{r2, false};  // :587-591
  if (to_system == false) {                              // :593-603 CALL_SYSTEM_ABI_REGISTERS
    for (uint32_t i = abi_first; i < abi_end; i++) local_state.registers[i] = PrimitiveValue::empty();
  } else {
    for (uint32_t i = abi_first; i < abi_end; i++) local_state.registers[i].is_pointer = false;
  }
  for (uint32_t i = res_first; i < res_end; i++) local_state.registers[i] = PrimitiveValue::empty();  // CALL_RESERVED_RANGE
  local_state.registers[CALL_IMPLICIT_PARAMETER_REG_IDX] = PrimitiveValue::empty();                  // :609-610
}

// ret.rs:9-265
void Vm::ret(const Decoded& op, const PreState& ps) {
  const zkw_isa_consts& K = isa->consts;
  uint8_t inner_variant = op.variant.variant;
  local_state.flags.reset();  // :27
  U256 src0 = ps.src0.value;
  bool src0_is_ptr = ps.src0.is_pointer;
  if (inner_variant == ZKW_RET_PANIC) { src0 = U256::zero(); src0_is_ptr = false; }  // :35-41
  FatPointer ptr = FatPointer::from_u256(src0);                                      // RetABI::from_u256
  int page_forwarding_mode = forward_type_from_u8(isa->consts.forwarding_codes, (uint8_t)(src0.l[3] >> 32));
  bool is_to_label = op.variant.flags & 1;  // RET_TO_LABEL_BIT_IDX
  uint16_t label_pc = op.imm_0;
  const CallStackEntry cs = local_state.callstack.current;
  uint32_t pointer_validation_exceptions = 0;
  if (cs.is_local_frame == false) {  // :58-96
    if (page_forwarding_mode == FWD_FORWARD_FAT_POINTER) {
      if (src0_is_ptr == false) inner_variant = ZKW_RET_PANIC;
      if (ptr.memory_page < cs.base_memory_page) inner_variant = ZKW_RET_PANIC;
    }
    bool validate_as_fresh = page_forwarding_mode != FWD_FORWARD_FAT_POINTER;
    pointer_validation_exceptions = fat_pointer_validate(ptr, validate_as_fresh);
    if (pointer_validation_exceptions != 0) inner_variant = ZKW_RET_PANIC;
    if (ptr.validate_as_slice() == false) inner_variant = ZKW_RET_PANIC;
    if (inner_variant == ZKW_RET_PANIC) ptr = FatPointer::empty();
  }
  uint32_t ergs_remaining = cs.ergs_remaining;
  bool has_returndata_ptr = !cs.is_local_frame;
  if (has_returndata_ptr) {  // :101-190
    if (inner_variant == ZKW_RET_OK || inner_variant == ZKW_RET_REVERT) {
      switch (page_forwarding_mode) {
        case FWD_FORWARD_FAT_POINTER: {
          uint32_t new_start = ptr.start + ptr.offset;
          uint32_t new_length = ptr.length - ptr.offset;
          ptr.start = new_start; ptr.length = new_length; ptr.offset = 0;
          break;
        }
        case FWD_USE_HEAP: ptr.memory_page = CallStackEntry::heap_page_from_base(cs.base_memory_page); break;
        default: ptr.memory_page = CallStackEntry::aux_heap_page_from_base(cs.base_memory_page); break;
      }
    }
    uint32_t memory_growth_in_bytes = 0;
    if (page_forwarding_mode == FWD_USE_HEAP || page_forwarding_mode == FWD_USE_AUX_HEAP) {  // :146-181
      uint32_t upper_bound = ptr.start + ptr.length;  // validated (or zeroed) pointer: no overflow
      if (pointer_validation_exceptions & FPV_DEREF_BEYOND_HEAP_RANGE) upper_bound = 0xffffffffu;
      uint32_t current_bound = page_forwarding_mode == FWD_USE_HEAP ? cs.heap_bound : cs.aux_heap_bound;
      memory_growth_in_bytes = upper_bound < current_bound ? 0 : upper_bound - current_bound;
    }
    uint32_t cost_of_memory_growth = memory_growth_in_bytes * K.memory_growth_ergs_per_byte;
    if (ergs_remaining >= cost_of_memory_growth) {
      ergs_remaining -= cost_of_memory_growth;
    } else {
      ergs_remaining = 0;
      inner_variant = ZKW_RET_PANIC;
      ptr = FatPointer::empty();
    }
  }
  bool panicked = inner_variant == ZKW_RET_REVERT || inner_variant == ZKW_RET_PANIC;  // :196
  CallStackEntry finished = finish_frame(local_state.monotonic_cycle_counter, panicked);  // :198-199
  is_to_label = is_to_label & finished.is_local_frame;                                   // :202
  if (finished.is_local_frame == false) {  // :204-236
    memory.finish_global_frame(finished.base_memory_page, ptr, local_state.timestamp);
    const uint32_t rr = isa->consts.ret_regs;  // RET_IMPLICIT_RETURNDATA_PARAMS_REGISTER, RET_RESERVED_REGISTER_0..2: table constants
    local_state.registers[rr & 0xff] = PrimitiveValue{ptr.to_u256(), true};
    local_state.registers[(rr >> 8) & 0xff] = PrimitiveValue::empty();
    local_state.registers[(rr >> 16) & 0xff] = PrimitiveValue::empty();
    local_state.registers[rr >> 24] = PrimitiveValue::empty();
    for (uint32_t i = (rr >> 24) + 1; i < ZKW_REGISTERS_COUNT; i++) local_state.registers[i] = PrimitiveValue::empty();  // :219-233 .skip(RET_RESERVED_REGISTER_2 + 1)
    local_state.context_u128_register[0] = local_state.context_u128_register[1] = 0;  // :236
  }
  CallStackEntry& next = local_state.callstack.current;
  next.ergs_remaining += ergs_remaining;  // :243
  if (is_to_label) next.pc = label_pc;
  else if (panicked) next.pc = finished.exception_handler_location;
  if (finished.is_local_frame == true) {  // :254-260
    REF_ASSERT(finished.heap_bound >= next.heap_bound, "heap bound shrank");
    REF_ASSERT(finished.aux_heap_bound >= next.aux_heap_bound, "aux heap bound shrank");
    next.heap_bound = finished.heap_bound;
    next.aux_heap_bound = finished.aux_heap_bound;
  }
  if (inner_variant == ZKW_RET_PANIC) local_state.flags.overflow_or_less_than_flag = true;  // :262-264
}

enum { UMA_INPUT_IS_NOT_POINTER_WHEN_EXPECTED = 1, UMA_DEREF_BEYOND_HEAP_RANGE = 2, UMA_OVERFLOW_ON_INCREMENT = 4, UMA_NOT_ENOUGH_ERGS_TO_GROW_MEMORY = 8 };  // uma.rs:11-17

// uma.rs:26-425
void Vm::uma(const Decoded& op, const PreState& ps) {
  const zkw_isa_consts& K = isa->consts;
  REF_ASSERT(!ps.has_dst0_mem, "UMA opcode has dst0 not in register");  // :45-48 (debug_assert)
  uint8_t v = op.variant.variant;
  local_state.callstack.current.pc = ps.new_pc;
  bool increment_offset = op.variant.flags & 1;  // UMA_INCREMENT_FLAG_IDX
  U256 src0_value = ps.src0.value;
  bool src0_is_ptr = ps.src0.is_pointer;
  U256 src1 = ps.src1.value;
  FatPointer fat_ptr = FatPointer::from_u256(src0_value);
  uint32_t exceptions = 0;
  bool skip_fat_ptr_oob = false, skip_deref_beyond = false;
  bool is_ptr_read = v == ZKW_UMA_FAT_PTR_READ;
  if (is_ptr_read && src0_is_ptr == false) exceptions |= UMA_INPUT_IS_NOT_POINTER_WHEN_EXPECTED;  // :73-78
  uint8_t memory_type;
  uint32_t base = local_state.callstack.current.base_memory_page;
  if (v == ZKW_UMA_HEAP_READ || v == ZKW_UMA_HEAP_WRITE) { fat_ptr.memory_page = CallStackEntry::heap_page_from_base(base); memory_type = ZKW_MEM_HEAP; }
  else if (v == ZKW_UMA_AUX_READ || v == ZKW_UMA_AUX_WRITE) { fat_ptr.memory_page = CallStackEntry::aux_heap_page_from_base(base); memory_type = ZKW_MEM_AUX_HEAP; }
  else memory_type = ZKW_MEM_FAT_PTR;
  uint32_t src_offset;
  if (is_ptr_read) {  // :110-120
    if (fat_ptr.validate_in_bounds() == false) skip_fat_ptr_oob = true;
    src_offset = fat_ptr.start + fat_ptr.offset;
  } else {  // :121-135
    U256 max_off = U256::from_u64(K.max_offset_to_deref_low);
    if (cmp(src0_value, max_off) > 0) { exceptions |= UMA_DEREF_BEYOND_HEAP_RANGE; skip_deref_beyond = true; }
    src_offset = fat_ptr.offset;
  }
  uint32_t incremented_offset = fat_ptr.offset + 32;
  bool increment_offset_of = incremented_offset < fat_ptr.offset;
  if (increment_offset_of) {  // :139-147
    exceptions |= UMA_OVERFLOW_ON_INCREMENT;
    if (!is_ptr_read) REF_ASSERT(exceptions & UMA_DEREF_BEYOND_HEAP_RANGE, "overflow on increment without deref-beyond");
  }
  CallStackEntry& cm = local_state.callstack.current;
  uint32_t memory_growth_in_bytes = 0;  // :152-194
  if (!is_ptr_read) {
    bool is_heap = v == ZKW_UMA_HEAP_READ || v == ZKW_UMA_HEAP_WRITE;
    uint32_t current_bound = is_heap ? cm.heap_bound : cm.aux_heap_bound;
    uint32_t upper_bound = incremented_offset;
    if (upper_bound < current_bound) {
      memory_growth_in_bytes = 0;
    } else {
      memory_growth_in_bytes = upper_bound - current_bound;
      if (is_heap) cm.heap_bound = upper_bound; else cm.aux_heap_bound = upper_bound;
    }
  }
  uint32_t cost_of_memory_growth = memory_growth_in_bytes * K.memory_growth_ergs_per_byte;  // :196-197
  if (exceptions & UMA_DEREF_BEYOND_HEAP_RANGE) cost_of_memory_growth = 0xffffffffu;        // :202-207
  uint32_t ergs_after;
  if (cm.ergs_remaining < cost_of_memory_growth) { ergs_after = 0; exceptions |= UMA_NOT_ENOUGH_ERGS_TO_GROW_MEMORY; }
  else ergs_after = cm.ergs_remaining - cost_of_memory_growth;
  cm.ergs_remaining = ergs_after;  // :217
  bool set_panic = exceptions != 0;
  bool skip_memory_access = (skip_fat_ptr_oob || skip_deref_beyond) || set_panic;  // :223-228

  uint32_t word_0 = src_offset / 32;
  uint32_t word_1 = word_0 + 1;
  uint32_t unalignment = src_offset % 32;
  uint32_t word_0_lowest_bytes = 32 - unalignment;
  uint32_t word_1_highest_bytes = unalignment;
  bool is_unaligned = unalignment != 0;
  MemoryLocation w0{memory_type, fat_ptr.memory_page, word_0}, w1{memory_type, fat_ptr.memory_page, word_1};
  uint32_t ts_read = timestamp_for_code_or_src_read(), ts_write = timestamp_for_dst_write();
  uint32_t cc = local_state.monotonic_cycle_counter;
  U256 word_0_read_value = skip_memory_access ? U256::zero() : read_memory(cc, ts_read, w0).value;                    // :265-274
  U256 word_1_read_value = (is_unaligned && !skip_memory_access) ? read_memory(cc, ts_read, w1).value : U256::zero();  // :276-288

  if (v == ZKW_UMA_HEAP_READ || v == ZKW_UMA_AUX_READ || v == ZKW_UMA_FAT_PTR_READ) {  // :291-348
    U256 result = shl(word_0_read_value, unalignment * 8);
    result = bit_or(result, shr(word_1_read_value, (32 - unalignment) * 8));
    if (v == ZKW_UMA_FAT_PTR_READ) {
      uint32_t bytes_beyond_the_bound = incremented_offset - fat_ptr.length;
      bool uf = incremented_offset < fat_ptr.length;
      if (uf || skip_memory_access) bytes_beyond_the_bound = 0;
      bytes_beyond_the_bound = bytes_beyond_the_bound % 32;
      result = shr(result, bytes_beyond_the_bound * 8);
      result = shl(result, bytes_beyond_the_bound * 8);
    }
    if (set_panic == false) {
      perform_dst0_update(cc, PrimitiveValue{result, false}, ps, op);
      if (increment_offset) {
        U256 updated = src0_value;
        updated.l[0] = (updated.l[0] & 0xffffffff00000000ULL) + (uint64_t)incremented_offset;
        perform_dst1_update(PrimitiveValue{updated, src0_is_ptr}, op.dst1_reg_idx);
      }
    } else {
      set_shorthand_panic();
    }
  } else {  // :349-423
    U256 new_word_0_value = shl(shr(word_0_read_value, word_0_lowest_bytes * 8), word_0_lowest_bytes * 8);
    new_word_0_value = bit_or(new_word_0_value, shr(src1, unalignment * 8));
    U256 new_word_1_value = shr(shl(word_1_read_value, word_1_highest_bytes * 8), word_1_highest_bytes * 8);
    new_word_1_value = bit_or(new_word_1_value, shl(src1, (32 - word_1_highest_bytes) * 8));
    if (skip_memory_access == false) write_memory(cc, ts_write, w0, PrimitiveValue{new_word_0_value, false});
    if (is_unaligned && skip_memory_access == false) write_memory(cc, ts_write, w1, PrimitiveValue{new_word_1_value, false});
    if (set_panic == false) {
      if (increment_offset) {
        U256 updated = src0_value;
        updated.l[0] = (updated.l[0] & 0xffffffff00000000ULL) + (uint64_t)incremented_offset;
        perform_dst0_update(cc, PrimitiveValue{updated, false}, ps, op);
      }
    } else {
      set_shorthand_panic();
    }
  }
}

// ---------------------------------------------------------------------------------------
// precompiles — zk_evm_abstractions::precompiles::{keccak256,sha256} @ v1.4.1 (absent crate).
// Restated from the published implementation as recalled; pinned only through the reference's
// keccak256 tests (digest + output placement).  The per-round read pattern is PARITY UNPINNED.
// ---------------------------------------------------------------------------------------
struct PrecompileCallABI {
  uint32_t input_memory_offset, input_memory_length, output_memory_offset, output_memory_length, memory_page_to_read, memory_page_to_write;
  uint64_t precompile_interpreted_data;
  static PrecompileCallABI from_u256(const U256& v) {
    return PrecompileCallABI{(uint32_t)v.l[0], (uint32_t)(v.l[0] >> 32), (uint32_t)v.l[1], (uint32_t)(v.l[1] >> 32), (uint32_t)v.l[2], (uint32_t)(v.l[2] >> 32), v.l[3]};
  }
};

void keccak256_rounds_function(uint32_t cc, const LogQuery& params, SimpleMemory& memory, std::vector<MemoryQuery>& reads, std::vector<MemoryQuery>& writes,
                               std::vector<std::pair<uint32_t, uint32_t>>& rounds) {
  const size_t KECCAK_RATE_BYTES = 136, MEMORY_READS_PER_CYCLE = 6, BUFFER_SIZE = MEMORY_READS_PER_CYCLE * 32;
  PrecompileCallABI abi = PrecompileCallABI::from_u256(params.key);
  uint32_t timestamp_to_read = params.timestamp;
  uint32_t timestamp_to_write = timestamp_to_read + 1;
  size_t input_byte_offset = abi.input_memory_offset;
  size_t bytes_left = abi.input_memory_length;
  size_t num_rounds = (bytes_left + (KECCAK_RATE_BYTES - 1)) / KECCAK_RATE_BYTES;
  size_t padding_space = bytes_left % KECCAK_RATE_BYTES;
  bool needs_extra_padding_round = padding_space == 0;
  if (needs_extra_padding_round) num_rounds += 1;
  uint8_t buffer[BUFFER_SIZE];
  std::memset(buffer, 0, sizeof buffer);
  size_t filled = 0;
  uint64_t state[25];
  std::memset(state, 0, sizeof state);
  for (size_t round = 0; round < num_rounds; round++) {
    bool is_last = round == num_rounds - 1;
    bool paddings_round = needs_extra_padding_round && is_last;
    const size_t reads_before = reads.size(), writes_before = writes.size();
    uint8_t bytes32_buffer[32];
    std::memset(bytes32_buffer, 0, 32);
    for (size_t idx = 0; idx < MEMORY_READS_PER_CYCLE; idx++) {
      size_t memory_index = input_byte_offset / 32, unalignment = input_byte_offset % 32;
      size_t at_most = 32 - unalignment;
      size_t meaningful = bytes_left >= at_most ? at_most : bytes_left;
      bool enough_buffer_space = filled + meaningful <= BUFFER_SIZE;
      bool nothing_to_read = meaningful == 0;
      bool should_read = !nothing_to_read && !paddings_round && enough_buffer_space;
      size_t bytes_to_fill = should_read ? meaningful : 0;
      if (should_read) {
        input_byte_offset += meaningful;
        bytes_left -= meaningful;
        MemoryQuery q{timestamp_to_read, MemoryLocation{ZKW_MEM_FAT_PTR, abi.memory_page_to_read, (uint32_t)memory_index}, U256::zero(), false, false};
        q = memory.execute_partial_query(cc, q);
        reads.push_back(q);
        to_big_endian(q.value, bytes32_buffer);
      }
      std::memcpy(buffer + filled, bytes32_buffer + unalignment, bytes_to_fill);
      filled += bytes_to_fill;
    }
    uint8_t block[KECCAK_RATE_BYTES];
    std::memcpy(block, buffer, KECCAK_RATE_BYTES);
    {  // ByteBuffer::consume
      uint8_t nb[BUFFER_SIZE];
      std::memset(nb, 0, sizeof nb);
      std::memcpy(nb, buffer + KECCAK_RATE_BYTES, BUFFER_SIZE - KECCAK_RATE_BYTES);
      std::memcpy(buffer, nb, BUFFER_SIZE);
      filled = filled < KECCAK_RATE_BYTES ? 0 : filled - KECCAK_RATE_BYTES;
    }
    if (paddings_round) {
      std::memset(block, 0, KECCAK_RATE_BYTES);
      block[0] = 0x01;
      block[KECCAK_RATE_BYTES - 1] = 0x80;
    } else if (is_last) {
      if (padding_space == KECCAK_RATE_BYTES - 1) {
        block[KECCAK_RATE_BYTES - 1] = 0x81;
      } else {
        block[padding_space] = 0x01;
        block[KECCAK_RATE_BYTES - 1] = 0x80;
      }
    }
    for (size_t i = 0; i < 17; i++) {
      uint64_t w = 0;
      for (int b = 0; b < 8; b++) w |= (uint64_t)block[i * 8 + b] << (8 * b);
      state[i] ^= w;
    }
    keccak_f1600(state);
    if (is_last) {
      uint8_t hash[32];
      for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) hash[i * 8 + b] = (uint8_t)(state[i] >> (8 * b));
      MemoryQuery w{timestamp_to_write, MemoryLocation{ZKW_MEM_HEAP, abi.memory_page_to_write, abi.output_memory_offset}, from_big_endian(hash), false, true};
      w = memory.execute_partial_query(cc, w);
      writes.push_back(w);
    }
    rounds.emplace_back((uint32_t)(reads.size() - reads_before), (uint32_t)(writes.size() - writes_before));
  }
}

// ecrecover precompile (absent crate zk_evm_abstractions; layout per the reference's test
// src/testing/tests/precompiles/ecrecover.rs:3-95): 4 reads @timestamp, 2 writes @timestamp + 1
void ecrecover_function(uint32_t cc, const LogQuery& params, SimpleMemory& memory, uint32_t layout, std::vector<MemoryQuery>& reads,
                        std::vector<MemoryQuery>& writes, std::vector<std::pair<uint32_t, uint32_t>>& rounds) {
  PrecompileCallABI abi = PrecompileCallABI::from_u256(params.key);
  uint32_t timestamp_to_read = params.timestamp, timestamp_to_write = timestamp_to_read + 1;
  U256 w[4];
  for (uint32_t k = 0; k < 4; k++) {
    MemoryQuery q{timestamp_to_read, MemoryLocation{ZKW_MEM_HEAP, abi.memory_page_to_read, abi.input_memory_offset + k}, U256::zero(), false, false};
    q = memory.execute_partial_query(cc, q);
    reads.push_back(q);
    w[k] = q.value;
  }
  const U256& vw = layout ? w[1] : w[3];
  const U256& r = layout ? w[2] : w[1];
  const U256& s = layout ? w[3] : w[2];
  REF_ASSERT(vw.l[0] <= 1 && vw.l[1] == 0 && vw.l[2] == 0 && vw.l[3] == 0, "ecrecover: v must be 0 or 1");
  uint8_t address[20];
  bool ok = secp::ecrecover(w[0], r, s, vw.l[0] == 1, address);
  uint8_t word[32] = {0};
  if (ok) std::memcpy(word + 12, address, 20);
  MemoryQuery m{timestamp_to_write, MemoryLocation{ZKW_MEM_HEAP, abi.memory_page_to_write, abi.output_memory_offset}, U256::from_u64(ok ? 1 : 0), false, true};
  m = memory.execute_partial_query(cc, m);
  writes.push_back(m);
  MemoryQuery a{timestamp_to_write, MemoryLocation{ZKW_MEM_HEAP, abi.memory_page_to_write, abi.output_memory_offset + 1}, from_big_endian(word), false, true};
  a = memory.execute_partial_query(cc, a);
  writes.push_back(a);
  rounds.emplace_back(4u, 2u);  // one round
}

void sha256_rounds_function(uint32_t cc, const LogQuery& params, SimpleMemory& memory, std::vector<MemoryQuery>& reads, std::vector<MemoryQuery>& writes,
                            std::vector<std::pair<uint32_t, uint32_t>>& rounds) {
  PrecompileCallABI abi = PrecompileCallABI::from_u256(params.key);
  uint32_t timestamp_to_read = params.timestamp;
  uint32_t timestamp_to_write = timestamp_to_read + 1;
  size_t num_rounds = (size_t)abi.precompile_interpreted_data;
  uint32_t current_read_offset = abi.input_memory_offset;
  uint32_t state[8];
  std::memcpy(state, SHA256_IV, sizeof state);
  for (size_t round = 0; round < num_rounds; round++) {
    uint8_t block[64];
    for (int qi = 0; qi < 2; qi++) {
      MemoryQuery q{timestamp_to_read, MemoryLocation{ZKW_MEM_HEAP, abi.memory_page_to_read, current_read_offset}, U256::zero(), false, false};
      q = memory.execute_partial_query(cc, q);
      current_read_offset += 1;
      reads.push_back(q);
      to_big_endian(q.value, block + qi * 32);
    }
    sha256_compress(state, block);
    if (round == num_rounds - 1) {
      uint8_t hash[32];
      for (int i = 0; i < 8; i++) {
        hash[4 * i] = (uint8_t)(state[i] >> 24); hash[4 * i + 1] = (uint8_t)(state[i] >> 16);
        hash[4 * i + 2] = (uint8_t)(state[i] >> 8); hash[4 * i + 3] = (uint8_t)state[i];
      }
      MemoryQuery w{timestamp_to_write, MemoryLocation{ZKW_MEM_HEAP, abi.memory_page_to_write, abi.output_memory_offset}, from_big_endian(hash), false, true};
      w = memory.execute_partial_query(cc, w);
      writes.push_back(w);
    }
    rounds.emplace_back(2u, round == num_rounds - 1 ? 1u : 0u);
  }
}

}  // namespace zko
