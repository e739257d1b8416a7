//! zkw-shim — `zk_evm` v1.4.1's caller-facing surface over libzkw.so (UNBUILT in the repository's image: no Rust
//! toolchain there; see Cargo.toml).
//!
//! The reference interprets one cycle per `VmState::cycle()` call (reference src/vm_state/cycle.rs:257) and calls its
//! `VmWitnessTracer` / `EventSink` from inside.  Here the cycles of a whole batch of instances are executed by the HIP
//! kernels first (`Batch::run`), and `BatchedVmState::cycle(&mut tracer)` then REPLAYS one cycle of one instance from
//! the finished trace: the same callbacks, with the same arguments, in the same order, and
//! `local_state` ends every call in the state the reference would be in.  A caller's loop
//!
//! ```ignore
//! let mut tracer = zk_evm::GenericNoopTracer::<SimpleMemory>::new();   // utils.rs:51-60
//! while !vm.execution_has_ended() { vm.cycle(&mut tracer)?; }          // cycle.rs:257-260
//! ```
//!
//! is unchanged: `cycle` keeps the reference's debug-`Tracer` parameter.  The four hooks of that trait are compiled out
//! unless the tracer's `CALL_*` constants say otherwise (tracing.rs:43-46); a tracer that asks for them is refused
//! loudly (they would need the decode internals of every cycle on the host — out of scope).  Oracles whose answers the VM consumes (Memory, Storage, Decommitter, Precompiles) are snapshotted
//! into the batch before the run (`Batch::set_*`: the reference impls `SimpleMemory`, `InMemoryStorage`,
//! `SimpleDecommitter` map one to one); oracles that only receive data (witness tracer, event sink) are driven by the
//! replay.  The C++ form of this file — era-zk_evm_amd/host/zk_evm.hpp — is what the repository's tests exercise.
pub mod ffi;

use std::ffi::CStr;

use ffi::*;
use zk_evm::aux_structures::{DecommittmentQuery, LogQuery, MemoryIndex, MemoryLocation, MemoryPage, MemoryQuery, Timestamp};
use zk_evm::ethereum_types::{Address, U256};
use zk_evm::flags::Flags;
use zk_evm::reference_impls::event_sink::EventMessage;
use zk_evm::tracing::Tracer;
use zk_evm::vm_state::{CallStackEntry, Callstack, PrimitiveValue, VmLocalState};
use zk_evm::witness_trace::VmWitnessTracer;
use zk_evm::zk_evm_abstractions::vm::{EventSink, MemoryType, PrecompileCyclesWitness, RefundType};
use zk_evm::zkevm_opcode_defs as defs;
use zk_evm::zkevm_opcode_defs::decoding::encoding_mode_production::EncodingModeProduction as E;

// ---------------------------------------------------------------------------------------------------------------
// conversions (include/zkw.h: U256 = 4 little-endian u64 limbs; addresses = little-endian bytes of the 160-bit integer)
// ---------------------------------------------------------------------------------------------------------------
pub fn u256_to_c(v: &U256) -> zkw_u256 {
    zkw_u256 { l: v.0 }
}
pub fn u256_from_c(v: &zkw_u256) -> U256 {
    U256(v.l)
}
pub fn address_from_c(b: &[u8; 20]) -> Address {
    let mut x = *b;
    x.reverse();
    Address::from(x)
}
/// the low 32 bits of an address (H160 is big-endian)
pub fn address_low_u32(a: &Address) -> u32 {
    let b = a.as_fixed_bytes();
    u32::from_be_bytes([b[16], b[17], b[18], b[19]])
}
pub fn address_to_c(a: &Address) -> [u8; 20] {
    let mut x = a.to_fixed_bytes();
    x.reverse();
    x
}
pub fn entry_from_c(e: &zkw_callstack_entry) -> CallStackEntry<8, E> {
    CallStackEntry {
        this_address: address_from_c(&e.this_address),
        msg_sender: address_from_c(&e.msg_sender),
        code_address: address_from_c(&e.code_address),
        base_memory_page: MemoryPage(e.base_memory_page),
        code_page: MemoryPage(e.code_page),
        sp: e.sp,
        pc: e.pc,
        exception_handler_location: e.exception_handler_location,
        ergs_remaining: e.ergs_remaining,
        this_shard_id: e.this_shard_id,
        caller_shard_id: e.caller_shard_id,
        code_shard_id: e.code_shard_id,
        is_static: e.is_static != 0,
        is_local_frame: e.is_local_frame != 0,
        context_u128_value: (e.context_u128_value[0] as u128) | ((e.context_u128_value[1] as u128) << 64),
        heap_bound: e.heap_bound,
        aux_heap_bound: e.aux_heap_bound,
    }
}
pub fn entry_to_c(e: &CallStackEntry<8, E>) -> zkw_callstack_entry {
    zkw_callstack_entry {
        this_address: address_to_c(&e.this_address),
        msg_sender: address_to_c(&e.msg_sender),
        code_address: address_to_c(&e.code_address),
        base_memory_page: e.base_memory_page.0,
        code_page: e.code_page.0,
        sp: e.sp,
        pc: e.pc,
        exception_handler_location: e.exception_handler_location,
        is_static: e.is_static as u8,
        is_local_frame: e.is_local_frame as u8,
        ergs_remaining: e.ergs_remaining,
        this_shard_id: e.this_shard_id,
        caller_shard_id: e.caller_shard_id,
        code_shard_id: e.code_shard_id,
        reserved0: 0,
        reserved1: 0,
        context_u128_value: [e.context_u128_value as u64, (e.context_u128_value >> 64) as u64],
        heap_bound: e.heap_bound,
        aux_heap_bound: e.aux_heap_bound,
    }
}
/// `VmLocalState` right after `push_bootloader_context` (reference helpers.rs:289-316) -> the C state + the inner entries
pub fn state_to_c(s: &VmLocalState<8, E>) -> (zkw_vm_local_state, Vec<zkw_callstack_entry>) {
    let mut bm = 0u16;
    let mut regs = [zkw_u256::default(); 15];
    for (i, r) in s.registers.iter().enumerate() {
        regs[i] = u256_to_c(&r.value);
        if r.is_pointer {
            bm |= 1 << i;
        }
    }
    let c = zkw_vm_local_state {
        previous_code_word: u256_to_c(&s.previous_code_word),
        registers: regs,
        register_ptr_bitmap: bm,
        flags: (s.flags.overflow_or_less_than_flag as u8) | ((s.flags.equality_flag as u8) << 1) | ((s.flags.greater_than_flag as u8) << 2),
        pending_exception: s.pending_exception as u8,
        previous_code_memory_page: s.previous_code_memory_page.0,
        timestamp: s.timestamp,
        monotonic_cycle_counter: s.monotonic_cycle_counter,
        spent_pubdata_counter: s.spent_pubdata_counter,
        memory_page_counter: s.memory_page_counter,
        absolute_execution_step: s.absolute_execution_step,
        current_ergs_per_pubdata_byte: s.current_ergs_per_pubdata_byte,
        tx_number_in_block: s.tx_number_in_block,
        previous_super_pc: s.previous_super_pc,
        callstack_depth: s.callstack.inner.len() as u32,
        context_u128_register: [s.context_u128_register as u64, (s.context_u128_register >> 64) as u64],
        current: entry_to_c(&s.callstack.current),
    };
    (c, s.callstack.inner.iter().map(entry_to_c).collect())
}

fn mem_query_from_c(q: &zkw_mem_query) -> MemoryQuery {
    let memory_type = match q.meta & 7 {
        0 => MemoryType::Stack,
        1 => MemoryType::Code,
        2 => MemoryType::Heap,
        3 => MemoryType::AuxHeap,
        _ => MemoryType::FatPointer,
    };
    MemoryQuery {
        timestamp: Timestamp(q.timestamp),
        location: MemoryLocation { memory_type, page: MemoryPage(q.page), index: MemoryIndex(q.index) },
        value: u256_from_c(&q.value),
        value_is_pointer: q.meta & 8 != 0,
        rw_flag: q.meta & 16 != 0,
    }
}
fn log_query_from_c(q: &zkw_log_query) -> LogQuery {
    LogQuery {
        timestamp: Timestamp(q.timestamp),
        tx_number_in_block: q.tx_number_in_block,
        aux_byte: q.aux_byte,
        shard_id: q.shard_id,
        address: address_from_c(&q.address),
        key: u256_from_c(&q.key),
        read_value: u256_from_c(&q.read_value),
        written_value: u256_from_c(&q.written_value),
        rw_flag: q.bools & 1 != 0,
        rollback: q.bools & 2 != 0,
        is_service: q.bools & 4 != 0,
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the ISA table from the REAL crate: this is what turns "parity given the same table" into drop-in parity
// (the same mapping as rust/zkw-refdump `dump-isa`; kept in one place there — see that file for the field notes)
// ---------------------------------------------------------------------------------------------------------------
pub fn isa_from_opcode_defs() -> Box<zkw_isa_table> {
    // SAFETY: plain-old-data struct, every field is written below or is a reserved zero
    let mut t: Box<zkw_isa_table> = unsafe { Box::new(std::mem::zeroed()) };
    const OPS: [&str; 16] = ["Invalid", "Nop", "Add", "Sub", "Mul", "Div", "Jump", "Context", "Shift", "Binop", "Ptr", "NearCall", "Log", "FarCall", "Ret", "UMA"];
    for idx in 0..2048usize {
        let v = &defs::OPCODES_TABLE[idx];
        let s = format!("{:?}", v.opcode); // "Add(Add)", "Log(StorageRead)", ...
        let (fam, inner) = match s.find('(') {
            Some(i) => (s[..i].to_string(), s[i + 1..s.len() - 1].to_string()),
            None => (s.clone(), String::new()),
        };
        let table: &[&str] = match fam.as_str() {
            "Context" => &["This", "Caller", "CodeAddress", "Meta", "ErgsLeft", "Sp", "GetContextU128", "SetContextU128", "SetErgsPerPubdataByte", "IncrementTxNumber"],
            "Shift" => &["Shl", "Shr", "Rol", "Ror"],
            "Binop" => &["Xor", "And", "Or"],
            "Ptr" => &["Add", "Sub", "Pack", "Shrink"],
            "Log" => &["StorageRead", "StorageWrite", "ToL1Message", "Event", "PrecompileCall"],
            "FarCall" => &["Normal", "Delegate", "Mimic"],
            "Ret" => &["Ok", "Revert", "Panic"],
            "UMA" => &["HeapRead", "HeapWrite", "AuxHeapRead", "AuxHeapWrite", "FatPointerRead"],
            _ => &[],
        };
        let mode = |o: &defs::Operand| -> u8 {
            use defs::{ImmMemHandlerFlags as F, Operand, RegOrImmFlags};
            match o {
                Operand::RegOnly | Operand::RegOrImm(RegOrImmFlags::UseRegOnly) | Operand::Full(F::UseRegOnly) => 0,
                Operand::Full(F::UseStackWithPushPop) => 1,
                Operand::Full(F::UseStackWithOffset) => 2,
                Operand::Full(F::UseAbsoluteOnStack) => 3,
                Operand::RegOrImm(RegOrImmFlags::UseImm16Only) | Operand::Full(F::UseImm16Only) => 4,
                Operand::Full(F::UseCodePage) => 5,
            }
        };
        let e = &mut t.entries[idx];
        e.opcode = OPS.iter().position(|n| *n == fam).unwrap_or(0) as u8;
        e.variant = table.iter().position(|n| *n == inner).unwrap_or(0) as u8;
        e.src0_mode = mode(&v.src0_operand_type);
        e.dst0_mode = mode(&v.dst0_operand_type);
        e.flags = v.flags.iter().enumerate().map(|(i, f)| (*f as u8) << i).sum();
        e.props = (v.is_explicit_panic() as u8)
            | ((v.requires_kernel_mode() as u8) << 1)
            | ((v.can_be_used_in_static_context() as u8) << 2)
            | ((v.swap_operands() as u8) << 3)
            | ((v.opcode.src0_can_be_pointer() as u8) << 4)
            | ((v.opcode.src1_can_be_pointer() as u8) << 5);
        e.price = defs::OPCODES_PRICES[idx] as u32;
    }
    use defs::decoding::VmEncodingMode;
    let c = &mut t.consts;
    c.nop_encoding = <E as VmEncodingMode<8>>::nop_encoding();
    c.exception_revert_encoding = <E as VmEncodingMode<8>>::exception_revert_encoding();
    c.panic_variant_idx = (c.exception_revert_encoding & 0x7ff) as u32;
    c.nop_variant_idx = (c.nop_encoding & 0x7ff) as u32;
    c.clip_mode = 1; // settled by zkw-refdump dump-isa from the crate's own from_u64_clipped; 1 = "lowest 16 bits" (jump.rs:23)
    c.time_delta_per_cycle = defs::TIME_DELTA_PER_CYCLE;
    c.new_memory_pages_per_far_call = defs::NEW_MEMORY_PAGES_PER_FAR_CALL;
    c.vm_max_stack_depth = defs::system_params::VM_MAX_STACK_DEPTH;
    c.initial_sp_on_far_call = defs::INITIAL_SP_ON_FAR_CALL as u32;
    c.new_frame_memory_stipend = defs::system_params::NEW_FRAME_MEMORY_STIPEND;
    c.memory_growth_ergs_per_byte = defs::system_params::MEMORY_GROWTH_ERGS_PER_BYTE;
    c.ergs_per_code_word_decommittment = defs::ERGS_PER_CODE_WORD_DECOMMITTMENT;
    c.initial_storage_write_pubdata_bytes = defs::system_params::INITIAL_STORAGE_WRITE_PUBDATA_BYTES as u32;
    c.l1_message_pubdata_bytes = defs::system_params::L1_MESSAGE_PUBDATA_BYTES;
    c.max_offset_to_deref_low = defs::uma::MAX_OFFSET_TO_DEREF.low_u32(); // the U256 bound of uma.rs:127
    c.deployer_address_low = address_low_u32(&defs::system_params::DEPLOYER_SYSTEM_CONTRACT_ADDRESS); // far_call.rs:6,136
    c.keccak_precompile_address = address_low_u32(&defs::system_params::KECCAK256_ROUND_FUNCTION_PRECOMPILE_FORMAL_ADDRESS) & 0xffff; // testing/tests/precompiles/keccak256.rs:114
    c.sha256_precompile_address = defs::system_params::SHA256_ROUND_FUNCTION_PRECOMPILE_ADDRESS as u32;
    c.ecrecover_precompile_address = defs::system_params::ECRECOVER_INNER_FUNCTION_PRECOMPILE_ADDRESS as u32;
    c.storage_aux_byte = defs::system_params::STORAGE_AUX_BYTE;
    c.event_aux_byte = defs::system_params::EVENT_AUX_BYTE;
    c.l1_message_aux_byte = defs::system_params::L1_MESSAGE_AUX_BYTE;
    c.precompile_aux_byte = defs::system_params::PRECOMPILE_AUX_BYTE;
    c.bootloader_calldata_page = defs::BOOTLOADER_CALLDATA_PAGE; // memory.rs:11
    // the conventions rounds 1-3 had compiled in: table constants since round 4 (include/zkw.h), filled from the real crate here
    c.call_regs = (defs::CALL_IMPLICIT_CALLDATA_FAT_PTR_REGISTER as u32)                 // far_call.rs:577
        | ((defs::CALL_IMPLICIT_CONSTRUCTOR_MARKER_REGISTER as u32) << 8)                 // far_call.rs:587
        | ((defs::CALL_IMPLICIT_PARAMETER_REG_IDX as u32) << 16);                         // far_call.rs:507,609
    c.call_ranges = (defs::CALL_SYSTEM_ABI_REGISTERS.start as u32)                        // far_call.rs:594
        | ((defs::CALL_SYSTEM_ABI_REGISTERS.end as u32) << 8)
        | ((defs::CALL_RESERVED_RANGE.start as u32) << 16)                                // far_call.rs:606
        | ((defs::CALL_RESERVED_RANGE.end as u32) << 24);
    c.ret_regs = (defs::RET_IMPLICIT_RETURNDATA_PARAMS_REGISTER as u32)                   // ret.rs:213
        | ((defs::RET_RESERVED_REGISTER_0 as u32) << 8)                                   // ret.rs:218-223
        | ((defs::RET_RESERVED_REGISTER_1 as u32) << 16)
        | ((defs::RET_RESERVED_REGISTER_2 as u32) << 24);
    c.forwarding_codes = (defs::FarCallForwardPageType::UseHeap as u32)                   // far_call.rs:255, ret.rs:59
        | ((defs::FarCallForwardPageType::ForwardFatPointer as u32) << 8)
        | ((defs::FarCallForwardPageType::UseAuxHeap as u32) << 16);
    c.unmapped_page = defs::UNMAPPED_PAGE;                                                // far_call.rs:9,162,439
    c.max_offset_for_add_sub = defs::ptr::MAX_OFFSET_FOR_ADD_SUB.low_u64();               // ptr.rs:47 (2^32)
    // the Condition each value of the 3-bit field names (cycle.rs:193-209), as its truth table over (lt_of | eq << 1 | gt << 2)
    c.condition_lut = 0;
    for field in 0..8u64 {
        let cond = defs::Condition::materialize_variant(field as u8);
        let mut row = 0u64;
        for f in 0..8u64 {
            let (lt, eq, gt) = (f & 1 != 0, f & 2 != 0, f & 4 != 0);
            let holds = match cond {
                defs::Condition::Always => true,
                defs::Condition::Gt => gt,
                defs::Condition::Lt => lt,
                defs::Condition::Eq => eq,
                defs::Condition::Ge => gt | eq,
                defs::Condition::Le => lt | eq,
                defs::Condition::Ne => !eq,
                defs::Condition::GtOrLt => gt | lt,
            };
            row |= (holds as u64) << f;
        }
        c.condition_lut |= row << (8 * field);
    }
    t
}

// ---------------------------------------------------------------------------------------------------------------
// context + batch: thin RAII wrappers over the C ABI
// ---------------------------------------------------------------------------------------------------------------
pub struct Context {
    pub raw: *mut zkw_ctx,
}
impl Context {
    pub fn new(device: i32) -> anyhow::Result<Self> {
        check_abi();
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { zkw_ctx_create(device, &mut raw) };
        anyhow::ensure!(rc == ZKW_OK, "zkw_ctx_create -> {}", rc);
        let ctx = Context { raw };
        let isa = isa_from_opcode_defs();
        ctx.check(unsafe { zkw_ctx_set_isa(raw, &*isa) }, "zkw_ctx_set_isa")?;
        Ok(ctx)
    }
    pub fn check(&self, rc: i32, what: &str) -> anyhow::Result<()> {
        if rc == ZKW_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(zkw_last_error(self.raw)) }.to_string_lossy().into_owned();
        anyhow::bail!("{} -> {}: {}", what, rc, msg)
    }
    /// zkw_ctx_set_option: `ZKW_OPT_STAGING_BUFFERS` (restaged heap images a held ticket may keep alive), `ZKW_OPT_LINK_FLAGS_OFF`
    /// (parts of the link format deliveries leave out: 1 = the values of memory reads travel again — cheaper for a host whose
    /// replay threads, not the link, are the bound), `ZKW_OPT_PACK_BLOCKS`
    pub fn set_option(&self, option: u32, value: u64) -> anyhow::Result<()> {
        self.check(unsafe { zkw_ctx_set_option(self.raw, option, value) }, "zkw_ctx_set_option")
    }
    /// `Blake2s256::digest` of the re-exported `zk_evm::blake2` (reference src/lib.rs:21) for a batch of messages,
    /// one message per GPU lane (zkw_blake2s256)
    pub fn blake2s256_batch(&self, messages: &[&[u8]]) -> anyhow::Result<Vec<[u8; 32]>> {
        let mut offsets = Vec::with_capacity(messages.len() + 1);
        let mut data = Vec::with_capacity(messages.iter().map(|m| m.len()).sum());
        offsets.push(0u64);
        for m in messages {
            data.extend_from_slice(m);
            offsets.push(data.len() as u64);
        }
        let mut out = vec![[0u8; 32]; messages.len()];
        let rc = unsafe { zkw_blake2s256(self.raw, data.as_ptr(), offsets.as_ptr(), messages.len() as u32, out.as_mut_ptr() as *mut u8) };
        self.check(rc, "zkw_blake2s256")?;
        Ok(out)
    }
}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe { zkw_ctx_destroy(self.raw) }
    }
}

/// N independent VM instances; the setters mirror the reference's `populate*` calls (entry-point table)
pub struct Batch<'a> {
    pub ctx: &'a Context,
    pub raw: *mut zkw_batch,
    pub n: u32,
    initial: Vec<(zkw_vm_local_state, Vec<zkw_callstack_entry>)>,
    blobs: Vec<Vec<U256>>,
}
impl<'a> Batch<'a> {
    pub fn new(ctx: &'a Context, n: u32, limits: &zkw_limits) -> anyhow::Result<Self> {
        let mut raw = std::ptr::null_mut();
        ctx.check(unsafe { zkw_batch_create(ctx.raw, n, limits, &mut raw) }, "zkw_batch_create")?;
        Ok(Batch { ctx, raw, n, initial: vec![], blobs: vec![vec![]] })
    }
    /// `SimpleMemory::populate_code` (memory.rs:271-284) + `SimpleDecommitter::populate` (decommitter.rs:23-28)
    pub fn add_code(&mut self, words: &[U256], hash: Option<U256>) -> anyhow::Result<u32> {
        let c: Vec<zkw_u256> = words.iter().map(u256_to_c).collect();
        let mut id = 0u32;
        self.ctx.check(unsafe { zkw_batch_add_code_blob(self.raw, c.as_ptr(), c.len() as u32, &mut id) }, "zkw_batch_add_code_blob")?;
        if let Some(h) = hash {
            self.ctx.check(unsafe { zkw_batch_add_decommit_preimage(self.raw, &u256_to_c(&h), id) }, "zkw_batch_add_decommit_preimage")?;
        }
        if self.blobs.len() <= id as usize {
            self.blobs.resize(id as usize + 1, vec![]);
        }
        self.blobs[id as usize] = words.to_vec();
        Ok(id)
    }
    pub fn set_code_page(&mut self, first: u32, count: u32, page: u32, blob: u32) -> anyhow::Result<()> {
        self.ctx.check(unsafe { zkw_batch_set_code_page(self.raw, first, count, page, blob) }, "zkw_batch_set_code_page")
    }
    /// the `VmLocalState` a caller would have handed to `VmState` (after `push_bootloader_context`)
    pub fn set_state(&mut self, instance: u32, state: &VmLocalState<8, E>) -> anyhow::Result<()> {
        let (c, inner) = state_to_c(state);
        self.ctx.check(unsafe { zkw_batch_set_state(self.raw, instance, 1, &c, inner.as_ptr(), inner.len() as u32) }, "zkw_batch_set_state")?;
        if self.initial.len() <= instance as usize {
            self.initial.resize(instance as usize + 1, (c, vec![]));
        }
        self.initial[instance as usize] = (c, inner);
        Ok(())
    }
    /// `SimpleMemory::populate_heap` (memory.rs:287-291)
    pub fn set_heap(&mut self, instance: u32, words: &[U256]) -> anyhow::Result<()> {
        let c: Vec<zkw_u256> = words.iter().map(u256_to_c).collect();
        self.ctx.check(unsafe { zkw_batch_set_heap(self.raw, instance, c.as_ptr(), c.len() as u32) }, "zkw_batch_set_heap")
    }
    /// `SimpleMemory::polulate_bootloaders_calldata` (memory.rs:293-298)
    pub fn set_bootloader_calldata(&mut self, instance: u32, words: &[U256]) -> anyhow::Result<()> {
        let c: Vec<zkw_u256> = words.iter().map(u256_to_c).collect();
        self.ctx.check(unsafe { zkw_batch_set_bootloader_calldata(self.raw, instance, c.as_ptr(), c.len() as u32) }, "zkw_batch_set_bootloader_calldata")
    }
    /// `vm.memory.dump_page_content_as_u256_words(page, range)` after the run (memory.rs:316-396; `VmState.memory` is a
    /// public field, vm_state/mod.rs:170)
    pub fn dump_page_content_as_u256_words(&self, instance: u32, page_number: u32, range: std::ops::Range<u32>) -> anyhow::Result<Vec<U256>> {
        let n = range.end.saturating_sub(range.start);
        let mut raw = vec![zkw_u256::default(); n as usize];
        self.ctx.check(unsafe { zkw_batch_get_page(self.raw, instance, page_number, range.start, n, raw.as_mut_ptr()) }, "zkw_batch_get_page")?;
        Ok(raw.iter().map(u256_from_c).collect())
    }
    /// `vm.memory.dump_page_content(page, range)` (memory.rs:300-314): big-endian words
    pub fn dump_page_content(&self, instance: u32, page_number: u32, range: std::ops::Range<u32>) -> anyhow::Result<Vec<[u8; 32]>> {
        Ok(self
            .dump_page_content_as_u256_words(instance, page_number, range)?
            .iter()
            .map(|w| {
                let mut b = [0u8; 32];
                w.to_big_endian(&mut b);
                b
            })
            .collect())
    }
    /// `InMemoryStorage::populate` (testing/storage.rs:26-31)
    pub fn set_storage(&mut self, instance: u32, elements: &[(u8, Address, U256, U256)]) -> anyhow::Result<()> {
        let c: Vec<zkw_storage_slot> = elements
            .iter()
            .map(|(shard, a, k, v)| zkw_storage_slot { key: u256_to_c(k), value: u256_to_c(v), address: address_to_c(a), shard_id: *shard, reserved0: [0; 3] })
            .collect();
        self.ctx.check(unsafe { zkw_batch_set_storage(self.raw, instance, c.as_ptr(), c.len() as u32) }, "zkw_batch_set_storage")
    }
    /// everything staged -> device; then every instance runs up to `max_cycles` cycles (the caller's cycle loop, for all)
    pub fn run(&mut self, max_cycles: u32) -> anyhow::Result<()> {
        self.ctx.check(unsafe { zkw_batch_upload(self.raw) }, "zkw_batch_upload")?;
        self.ctx.check(unsafe { zkw_batch_reset(self.raw, std::ptr::null_mut()) }, "zkw_batch_reset")?;
        self.ctx.check(unsafe { zkw_batch_run(self.raw, max_cycles, std::ptr::null_mut()) }, "zkw_batch_run")?;
        self.ctx.check(unsafe { zkw_batch_sync(self.raw) }, "zkw_batch_sync")
    }
    /// zkw_batches_step over several batches of one context: restore, run and commit them with fused launches (what a caller
    /// that owns many blocks / transactions does per scheduling quantum); `queue_mask`: bit 0 memory, 1 log, 2 decommit queue
    pub fn step_many(batches: &mut [&mut Batch<'a>], max_cycles: u32, queue_mask: u32) -> anyhow::Result<()> {
        let ctx = batches.first().map(|b| b.ctx).ok_or_else(|| anyhow::anyhow!("step_many: no batches"))?;
        for b in batches.iter() {
            ctx.check(unsafe { zkw_batch_upload(b.raw) }, "zkw_batch_upload")?;
        }
        let raw: Vec<*mut zkw_batch> = batches.iter().map(|b| b.raw).collect();
        ctx.check(unsafe { zkw_batches_step(raw.as_ptr(), raw.len() as u32, max_cycles, queue_mask, std::ptr::null_mut()) }, "zkw_batches_step")?;
        for b in batches.iter() {
            ctx.check(unsafe { zkw_batch_sync(b.raw) }, "zkw_batch_sync")?;
        }
        Ok(())
    }
    /// New VmLocalStates (and heap images) for every instance of an uploaded batch that keeps its geometry — what a caller that
    /// pushes the next transactions through the same batch object does instead of upload: VmState::empty_state +
    /// push_bootloader_context with other values (vm_state/mod.rs:188-207, helpers.rs:289-316), SimpleMemory::populate_heap
    /// (reference_impls/memory.rs:287-291).  Asynchronous on the library's side (zkw_batch_restage); the traces rebuilt afterwards
    /// replay onto these states.
    pub fn restage(&mut self, states: &[VmLocalState<8, E>], heaps: Option<&[Vec<U256>]>) -> anyhow::Result<()> {
        anyhow::ensure!(states.len() == self.n as usize, "restage: one state per instance");
        let mut cs = Vec::with_capacity(states.len());
        for (i, st) in states.iter().enumerate() {
            let (c, inner) = state_to_c(st);
            self.initial[i] = (c, inner);
            cs.push(c);
        }
        let (hp, nh, flat): (*const zkw_u256, u32, Vec<zkw_u256>) = match heaps {
            Some(h) => {
                anyhow::ensure!(h.len() == self.n as usize, "restage: one heap image per instance");
                let n = h.first().map(|v| v.len()).unwrap_or(0);
                // (the C ABI takes one flat [n_instances][n_words] array: images of different lengths would shift every later instance)
                anyhow::ensure!(h.iter().all(|v| v.len() == n), "restage: the heap images must all have the uploaded length");
                let flat: Vec<zkw_u256> = h.iter().flat_map(|v| v.iter().map(u256_to_c)).collect();
                (flat.as_ptr(), n as u32, flat)
            }
            None => (std::ptr::null(), 0, Vec::new()),
        };
        let rc = unsafe { zkw_batch_restage(self.raw, cs.as_ptr(), hp, nh, std::ptr::null_mut()) };
        drop(flat);
        self.ctx.check(rc, "zkw_batch_restage")
    }
    /// The zero-copy form of `restage`: `fill` writes the next inputs straight into the batch's pinned staging buffers
    /// (zkw_batch_staging: states [n_instances], heap images [n_instances][n_heap_words] — the slice is empty when the batch was
    /// uploaded without heaps) and the library copies nothing on the host.  `fill` returns whether it wrote heap images.
    pub fn restage_in_place<F>(&mut self, fill: F) -> anyhow::Result<()>
    where
        F: FnOnce(&mut [zkw_vm_local_state], &mut [zkw_u256], usize) -> bool,
    {
        let (mut sp, mut hp, mut nh): (*mut zkw_vm_local_state, *mut zkw_u256, u32) = (std::ptr::null_mut(), std::ptr::null_mut(), 0);
        self.ctx.check(unsafe { zkw_batch_staging(self.raw, &mut sp, &mut hp, &mut nh) }, "zkw_batch_staging")?;
        let n = self.n as usize;
        let states = unsafe { std::slice::from_raw_parts_mut(sp, n) };
        let heaps: &mut [zkw_u256] = if nh == 0 || hp.is_null() { &mut [] } else { unsafe { std::slice::from_raw_parts_mut(hp, n * nh as usize) } };
        let with_heaps = fill(states, heaps, nh as usize);
        for (i, c) in states.iter().enumerate() {
            self.initial[i].0 = *c;  // (the inner entries of the callstack keep the uploaded geometry: zkw_batch_restage checks it)
        }
        let (hw, nw) = if with_heaps && nh != 0 { (hp as *const zkw_u256, nh) } else { (std::ptr::null(), 0) };
        let rc = unsafe { zkw_batch_restage(self.raw, sp as *const zkw_vm_local_state, hw, nw, std::ptr::null_mut()) };
        self.ctx.check(rc, "zkw_batch_restage")
    }
    /// the queue commitments of every instance after a run: [instance][memory, log, decommit][4] Goldilocks elements
    /// (the build's own sponge spec: the reference has none, far_call.rs:29-32)
    pub fn commitments(&self, queue_mask: u32) -> anyhow::Result<Vec<[[u64; 4]; 3]>> {
        self.ctx.check(unsafe { zkw_batch_commit(self.raw, queue_mask, std::ptr::null_mut()) }, "zkw_batch_commit")?;
        let mut out = vec![[[0u64; 4]; 3]; self.n as usize];
        self.ctx.check(unsafe { zkw_batch_get_commitments(self.raw, out.as_mut_ptr() as *mut u64) }, "zkw_batch_get_commitments")?;
        Ok(out)
    }
    /// `get_final_net_states` of the reference's test tooling (testing/mod.rs:42-71) for one instance, netted on the device:
    /// (full_storage_access_history, storage_pre_shard, events_log_history, events, l1_messages) with the rollback entries
    /// of panicked frames spliced in as the reference's oracles do (testing/storage.rs:144-186, event_sink.rs:160-176)
    #[allow(clippy::type_complexity)]
    pub fn net_state(&self, instance: u32) -> anyhow::Result<(Vec<LogQuery>, Vec<(u8, Address, U256, U256)>, Vec<LogQuery>, Vec<EventMessage>, Vec<EventMessage>)> {
        let mut ns: zkw_net_state = unsafe { std::mem::zeroed() };
        self.ctx.check(unsafe { zkw_batch_get_net_state(self.raw, instance, &mut ns) }, "zkw_batch_get_net_state")?;
        let logs = |p: *const zkw_log_query, n: u32| -> Vec<LogQuery> { (0..n as usize).map(|i| log_query_from_c(unsafe { &*p.add(i) })).collect() };
        let msgs = |p: *const zkw_event_message, n: u32| -> Vec<EventMessage> {
            (0..n as usize)
                .map(|i| {
                    let m = unsafe { &*p.add(i) };
                    EventMessage { shard_id: m.shard_id, is_first: m.is_first != 0, tx_number_in_block: m.tx_number_in_block, address: address_from_c(&m.address), key: u256_from_c(&m.key), value: u256_from_c(&m.value) }
                })
                .collect()
        };
        let storage = (0..ns.n_final_storage as usize)
            .map(|i| {
                let s = unsafe { &*ns.final_storage.add(i) };
                (s.shard_id, address_from_c(&s.address), u256_from_c(&s.key), u256_from_c(&s.value))
            })
            .collect();
        Ok((logs(ns.storage_history, ns.n_storage_history), storage, logs(ns.event_history, ns.n_event_history), msgs(ns.events, ns.n_events), msgs(ns.l1_messages, ns.n_l1_messages)))
    }
    /// the drop-in `VmState` of one instance, served from the finished run
    pub fn vm_state<EV: EventSink, WT: VmWitnessTracer<8, E>>(&self, instance: u32, event_sink: EV, witness_tracer: WT) -> anyhow::Result<BatchedVmState<'_, EV, WT>> {
        let mut trace: zkw_instance_trace = unsafe { std::mem::zeroed() };
        self.ctx.check(unsafe { zkw_batch_get_instance_trace(self.raw, instance, &mut trace) }, "zkw_batch_get_instance_trace")?;
        let (c, inner) = &self.initial[instance as usize];
        Ok(BatchedVmState { local_state: local_state_from_c(c, inner), event_sink, witness_tracer, trace, k: 0, blobs: &self.blobs })
    }
}
impl<'a> Drop for Batch<'a> {
    fn drop(&mut self) {
        unsafe { zkw_batch_destroy(self.raw) }
    }
}

/// zkw_delivery: whole steps of groups of batches delivered into a persistent ring of pinned host slots by ONE pack kernel per
/// step (the kernel's stores are the transfer), for the consumer the reference names — a VmWitnessTracer on the HOST
/// (witness_trace/mod.rs:11-72).  `submit` behind a step, `wait`, then `vm_state` per instance (the drop-in BatchedVmState served
/// from the ring: nothing is read from the device any more) and `release`; step k's delivery runs beside step k + 1's kernels.
pub struct Delivery<'a> {
    ctx: &'a Context,
    raw: *mut zkw_delivery,
}
impl<'a> Delivery<'a> {
    pub fn new(ctx: &'a Context, n_slots: u32, slot_bytes: u64, host_threads: u32) -> anyhow::Result<Self> {
        let mut raw = std::ptr::null_mut();
        ctx.check(unsafe { zkw_delivery_create(ctx.raw, n_slots, slot_bytes, host_threads, &mut raw) }, "zkw_delivery_create")?;
        Ok(Delivery { ctx, raw })
    }
    /// upper bound of the block a step of these batches can produce (every stream at its capacity)
    pub fn worst_case_bytes(ctx: &Context, batches: &[&Batch<'a>]) -> anyhow::Result<u64> {
        let raw: Vec<*mut zkw_batch> = batches.iter().map(|b| b.raw).collect();
        let mut out = 0u64;
        ctx.check(unsafe { zkw_delivery_slot_bytes(raw.as_ptr(), raw.len() as u32, &mut out) }, "zkw_delivery_slot_bytes")?;
        Ok(out)
    }
    /// delivers the step these batches have just run (asynchronous); the ticket names it from now on
    pub fn submit(&mut self, batches: &[&Batch<'a>]) -> anyhow::Result<u32> {
        let raw: Vec<*mut zkw_batch> = batches.iter().map(|b| b.raw).collect();
        let mut ticket = 0u32;
        self.ctx.check(unsafe { zkw_delivery_submit(self.raw, raw.as_ptr(), raw.len() as u32, std::ptr::null_mut(), &mut ticket) }, "zkw_delivery_submit")?;
        Ok(ticket)
    }
    /// blocks until the block of `ticket` is in the ring: (bytes that crossed the link, device time of the pack kernel in ms)
    pub fn wait(&mut self, ticket: u32) -> anyhow::Result<(u64, f64)> {
        let mut info = zkw_delivered::default();
        self.ctx.check(unsafe { zkw_delivery_wait(self.raw, ticket, &mut info) }, "zkw_delivery_wait")?;
        Ok((info.bytes, info.pack_ms))
    }
    /// the drop-in `VmState` of one instance of a delivered step (Batch::vm_state, served from the ring)
    pub fn delivered_trace<'b, EV: EventSink, WT: VmWitnessTracer<8, E>>(&'b self, ticket: u32, batch_index: u32, batch: &'b Batch<'a>, instance: u32, event_sink: EV, witness_tracer: WT)
        -> anyhow::Result<BatchedVmState<'b, EV, WT>> {
        let mut trace: zkw_instance_trace = unsafe { std::mem::zeroed() };
        self.ctx.check(unsafe { zkw_delivery_get_instance_trace(self.raw, ticket, batch_index, instance, &mut trace) }, "zkw_delivery_get_instance_trace")?;
        let (c, inner) = &batch.initial[instance as usize];
        Ok(BatchedVmState { local_state: local_state_from_c(c, inner), event_sink, witness_tracer, trace, k: 0, blobs: &batch.blobs })
    }
    /// every (instance, cycle) of the step through the library's thread pool: the built-in consumer (cycles, checksum)
    pub fn replay(&mut self, ticket: u32) -> anyhow::Result<(u64, u64)> {
        let (mut n, mut sum) = (0u64, 0u64);
        self.ctx.check(unsafe { zkw_delivery_replay(self.raw, ticket, None, std::ptr::null_mut(), &mut n, &mut sum) }, "zkw_delivery_replay")?;
        Ok((n, sum))
    }
    pub fn release(&mut self, ticket: u32) -> anyhow::Result<()> {
        self.ctx.check(unsafe { zkw_delivery_release(self.raw, ticket) }, "zkw_delivery_release")
    }
}
impl<'a> Drop for Delivery<'a> {
    fn drop(&mut self) {
        unsafe { zkw_delivery_destroy(self.raw) }
    }
}

fn local_state_from_c(c: &zkw_vm_local_state, inner: &[zkw_callstack_entry]) -> VmLocalState<8, E> {
    let mut registers = [PrimitiveValue::empty(); 15];
    for i in 0..15 {
        registers[i] = PrimitiveValue { value: u256_from_c(&c.registers[i]), is_pointer: (c.register_ptr_bitmap >> i) & 1 != 0 };
    }
    VmLocalState {
        previous_code_word: u256_from_c(&c.previous_code_word),
        previous_code_memory_page: MemoryPage(c.previous_code_memory_page),
        registers,
        flags: Flags { overflow_or_less_than_flag: c.flags & 1 != 0, equality_flag: c.flags & 2 != 0, greater_than_flag: c.flags & 4 != 0 },
        timestamp: c.timestamp,
        monotonic_cycle_counter: c.monotonic_cycle_counter,
        spent_pubdata_counter: c.spent_pubdata_counter,
        memory_page_counter: c.memory_page_counter,
        absolute_execution_step: c.absolute_execution_step,
        current_ergs_per_pubdata_byte: c.current_ergs_per_pubdata_byte,
        tx_number_in_block: c.tx_number_in_block,
        pending_exception: c.pending_exception != 0,
        previous_super_pc: c.previous_super_pc,
        context_u128_register: (c.context_u128_register[0] as u128) | ((c.context_u128_register[1] as u128) << 64),
        callstack: Callstack { current: entry_from_c(&c.current), inner: inner.iter().map(entry_from_c).collect() },
    }
}

// ---------------------------------------------------------------------------------------------------------------
// BatchedVmState: VmState's surface, cycle() = replay
// ---------------------------------------------------------------------------------------------------------------
pub struct BatchedVmState<'b, EV: EventSink, WT: VmWitnessTracer<8, E>> {
    pub local_state: VmLocalState<8, E>,
    pub event_sink: EV,
    pub witness_tracer: WT,
    trace: zkw_instance_trace,
    k: u32,
    blobs: &'b [Vec<U256>],
}

impl<'b, EV: EventSink, WT: VmWitnessTracer<8, E>> BatchedVmState<'b, EV, WT> {
    pub fn execution_has_ended(&self) -> bool {
        self.local_state.execution_has_ended() // vm_state/mod.rs:214-216
    }

    /// `VmState::cycle` (cycle.rs:257-429, signature :257-260): Ok(()) per replayed cycle; `Err` where the reference
    /// returns `Err` (decommitter.rs:54-56) or where the recorded cycles are exhausted; panics where the reference panics.
    /// `tracer` is the reference's debug tracer: accepted so that the caller's loop compiles unchanged, never called —
    /// its hooks are const-gated (tracing.rs:43-46) and a tracer that enables one is refused.
    pub fn cycle<DT: Tracer<8, E>>(&mut self, _tracer: &mut DT) -> anyhow::Result<()> {
        assert!(
            !(DT::CALL_BEFORE_DECODING || DT::CALL_AFTER_DECODING || DT::CALL_BEFORE_EXECUTION || DT::CALL_AFTER_EXECUTION),
            "zkw-shim replays finished cycles: the debug Tracer hooks (tracing.rs:40-72) are not available"
        );
        let t = &self.trace;
        if self.k >= t.n_cycles {
            match t.status {
                ZKW_STATUS_UNKNOWN_CODE_HASH => anyhow::bail!("Code hash must be known"),
                ZKW_STATUS_REFERENCE_PANIC => panic!("the reference panics in this cycle (assert / unwrap / unreachable)"),
                ZKW_STATUS_LIMIT => anyhow::bail!("libzkw: a batch capacity (zkw_limits) was exceeded in this cycle"),
                _ => anyhow::bail!("libzkw: no more recorded cycles (run the batch further)"),
            }
        }
        let k = self.k as usize;
        let cc = self.local_state.monotonic_cycle_counter;
        self.witness_tracer.start_new_execution_cycle(&self.local_state); // cycle.rs:34
        let pre = self.local_state.callstack.current;
        let fetched = !self.local_state.pending_exception
            && (pre.code_page != self.local_state.previous_code_memory_page || (pre.pc >> 2) != self.local_state.previous_super_pc); // cycle.rs:58-60
        let (mem, log, aux, mo, lo, ao) = unsafe {
            (
                std::slice::from_raw_parts(t.mem, t.n_mem as usize),
                std::slice::from_raw_parts(t.log, t.n_log as usize),
                std::slice::from_raw_parts(t.aux, t.n_aux as usize),
                std::slice::from_raw_parts(t.mem_off, t.n_cycles as usize + 1),
                std::slice::from_raw_parts(t.log_off, t.n_cycles as usize + 1),
                std::slice::from_raw_parts(t.aux_off, t.n_cycles as usize + 1),
            )
        };
        let (mut mi, me, mut li, le, mut ai, ae) = (mo[k] as usize, mo[k + 1] as usize, lo[k] as usize, lo[k + 1] as usize, ao[k] as usize, ao[k + 1] as usize);
        let mut first_mem = true;
        let mut precompile: Option<LogQuery> = None;
        let (mut pin, mut pout): (Vec<MemoryQuery>, Vec<MemoryQuery>) = (vec![], vec![]);
        let mut cold: Option<zkw_aux_event> = None;
        // merge the three streams of this cycle by their in-cycle sequence number (ties only at the saturated value 255:
        // memory, then log, then aux)
        while mi < me || li < le || ai < ae {
            let ms = if mi < me { mem[mi].seq as u32 } else { u32::MAX };
            let ls = if li < le { log[li].seq as u32 } else { u32::MAX };
            let xs = if ai < ae { aux[ai].seq as u32 } else { u32::MAX };
            if ms <= ls && ms <= xs {
                let r = &mem[mi];
                mi += 1;
                let q = mem_query_from_c(r);
                match r.meta >> 5 {
                    1 => pin.push(q),
                    2 => pout.push(q),
                    _ => {
                        self.flush_precompile(cc, &mut precompile, &mut pin, &mut pout);
                        if first_mem && fetched {
                            self.local_state.previous_code_word = q.value; // cycle.rs:83
                        }
                        self.witness_tracer.add_memory_query(cc, q); // helpers.rs:34-37
                    }
                }
                first_mem = false;
            } else if ls <= xs {
                self.flush_precompile(cc, &mut precompile, &mut pin, &mut pout);
                let r = &log[li];
                li += 1;
                let q = log_query_from_c(r);
                if r.kind == 1 {
                    self.witness_tracer.record_refund_for_query(cc, q, RefundType::None); // helpers.rs:128-132 (InMemoryStorage refunds nothing)
                } else {
                    if q.aux_byte == defs::system_params::EVENT_AUX_BYTE || q.aux_byte == defs::system_params::L1_MESSAGE_AUX_BYTE {
                        self.event_sink.add_partial_query(cc, q); // helpers.rs:157-162
                    }
                    self.witness_tracer.add_log_query(cc, q);
                    if q.aux_byte == defs::system_params::PRECOMPILE_AUX_BYTE {
                        precompile = Some(q); // helpers.rs:207-222
                    }
                }
            } else {
                self.flush_precompile(cc, &mut precompile, &mut pin, &mut pout);
                let e = aux[ai];
                ai += 1;
                match e.r#type {
                    1 => {
                        // helpers.rs:225-246
                        let (prev, next) = unsafe { (entry_from_c(&e.u.frame.previous), entry_from_c(&e.u.frame.next)) };
                        self.event_sink.start_frame(Timestamp(self.local_state.timestamp));
                        self.witness_tracer.start_new_execution_context(cc, &prev, &next);
                        self.local_state.callstack.inner.push(prev);
                        self.local_state.callstack.current = next;
                    }
                    2 => {
                        // helpers.rs:248-264
                        self.event_sink.finish_frame(e.flag != 0, Timestamp(self.local_state.timestamp));
                        self.witness_tracer.finish_execution_context(cc, e.flag != 0);
                        self.local_state.callstack.current = self.local_state.callstack.inner.pop().expect("frame finish on an empty callstack");
                    }
                    3 => {
                        // helpers.rs:164-194
                        let q = DecommittmentQuery {
                            hash: u256_from_c(unsafe { &e.u.hash }),
                            timestamp: Timestamp(e.a),
                            memory_page: MemoryPage(e.b),
                            decommitted_length: (e.c & 0xffff) as u16,
                            is_fresh: e.flag != 0,
                        };
                        let words = if q.is_fresh { self.blobs.get((e.c >> 16) as usize).cloned().unwrap_or_default() } else { vec![] };
                        self.witness_tracer.add_decommittment(cc, q, words);
                    }
                    _ => cold = Some(e),
                }
            }
        }
        self.flush_precompile(cc, &mut precompile, &mut pin, &mut pout);
        // the state after the cycle: the CycleRecord + what the events above changed
        let rec = unsafe { &*t.records.add(k) };
        let s = &mut self.local_state;
        for i in 0..15 {
            s.registers[i] = PrimitiveValue { value: u256_from_c(&rec.registers[i]), is_pointer: (rec.tail.register_ptr_bitmap >> i) & 1 != 0 };
        }
        s.flags = Flags { overflow_or_less_than_flag: rec.tail.flags & 1 != 0, equality_flag: rec.tail.flags & 2 != 0, greater_than_flag: rec.tail.flags & 4 != 0 };
        s.pending_exception = rec.tail.flags & 8 != 0;
        s.timestamp = rec.tail.timestamp;
        s.previous_super_pc = rec.tail.previous_super_pc;
        s.previous_code_memory_page = pre.code_page; // cycle.rs:49
        s.monotonic_cycle_counter = cc + 1; // cycle.rs:411
        assert_eq!(s.callstack.inner.len(), rec.tail.callstack_depth as usize, "replay: callstack depth mismatch");
        let cur = &mut s.callstack.current;
        cur.pc = rec.tail.pc;
        cur.sp = rec.tail.sp;
        cur.ergs_remaining = rec.tail.ergs_remaining;
        cur.heap_bound = rec.tail.heap_bound;
        cur.aux_heap_bound = rec.tail.aux_heap_bound;
        if let Some(e) = cold {
            s.spent_pubdata_counter = e.a;
            s.current_ergs_per_pubdata_byte = e.b;
            s.tx_number_in_block = e.c as u16;
            let c = unsafe { e.u.cold };
            s.context_u128_register = (c.context_u128_register[0] as u128) | ((c.context_u128_register[1] as u128) << 64);
            s.memory_page_counter = c.memory_page_counter;
        }
        self.witness_tracer.end_execution_cycle(&self.local_state); // cycle.rs:413
        self.k += 1;
        Ok(())
    }

    /// `add_precompile_call_result` for every call to a known precompile — also one with no rounds (helpers.rs:210-221)
    fn flush_precompile(&mut self, cc: u32, call: &mut Option<LogQuery>, pin: &mut Vec<MemoryQuery>, pout: &mut Vec<MemoryQuery>) {
        if let Some(q) = call.take() {
            if let Some(rounds) = round_witness(&q, pin, pout) {
                self.witness_tracer.add_precompile_call_result(cc, q, std::mem::take(pin), std::mem::take(pout), rounds);
            }
        }
        pin.clear();
        pout.clear();
    }
}

/// Rebuilds `PrecompileCyclesWitness` from the ordered reads / writes of a call: the rounds consume them in order
/// (sha256: two reads per round, the write in the last; ecrecover: one round; keccak256: what each 136-byte block still
/// lacks).  The exact per-round structs live in `zk_evm_abstractions::precompiles::{sha256, keccak256, ecrecover}`;
/// the grouping logic is the one of era-zk_evm_amd/host/zk_evm.hpp `precompile_round_witness` (tested there).
fn round_witness(call: &LogQuery, pin: &[MemoryQuery], pout: &[MemoryQuery]) -> Option<PrecompileCyclesWitness> {
    use zk_evm::zk_evm_abstractions::precompiles::{ecrecover::ECRecoverRoundWitness, keccak256::Keccak256RoundWitness, sha256::Sha256RoundWitness};
    let low = {
        let b = call.address.to_fixed_bytes();
        u16::from_be_bytes([b[18], b[19]])
    };
    if low as u64 == defs::system_params::SHA256_ROUND_FUNCTION_PRECOMPILE_ADDRESS as u64 { // (cast as in testing/tests/precompiles/sha256.rs:91)
        let rounds = call.key.0[3] as usize;
        let mut v = vec![];
        for r in 0..rounds {
            v.push(Sha256RoundWitness {
                new_request: if r == 0 { Some(*call) } else { None },
                reads: [pin[2 * r], pin[2 * r + 1]],
                writes: if r + 1 == rounds { Some([pout[0]]) } else { None },
            });
        }
        Some(PrecompileCyclesWitness::Sha256(v))
    } else if low as u64 == defs::system_params::ECRECOVER_INNER_FUNCTION_PRECOMPILE_ADDRESS as u64 {
        Some(PrecompileCyclesWitness::ECRecover(vec![ECRecoverRoundWitness { new_request: *call, reads: [pin[0], pin[1], pin[2], pin[3]], writes: [pout[0], pout[1]] }]))
    } else if low as u32 == address_low_u32(&defs::system_params::KECCAK256_ROUND_FUNCTION_PRECOMPILE_FORMAL_ADDRESS) & 0xffff {
        const RATE: usize = 136;
        const PER_CYCLE: usize = 6;
        const BUF: usize = PER_CYCLE * 32;
        let (mut offset, mut left) = ((call.key.0[0] & 0xffff_ffff) as usize, (call.key.0[0] >> 32) as usize);
        let mut rounds = (left + RATE - 1) / RATE;
        let extra = left % RATE == 0;
        if extra {
            rounds += 1;
        }
        let (mut filled, mut ri) = (0usize, 0usize);
        let mut v = vec![];
        for r in 0..rounds {
            let last = r + 1 == rounds;
            let mut reads: [Option<MemoryQuery>; PER_CYCLE] = [None; PER_CYCLE];
            for slot in reads.iter_mut() {
                let at_most = 32 - offset % 32;
                let meaningful = left.min(at_most);
                if meaningful != 0 && !(extra && last) && filled + meaningful <= BUF {
                    offset += meaningful;
                    left -= meaningful;
                    filled += meaningful;
                    *slot = Some(pin[ri]);
                    ri += 1;
                }
            }
            filled = filled.saturating_sub(RATE);
            v.push(Keccak256RoundWitness { new_request: if r == 0 { Some(*call) } else { None }, reads, writes: if last { Some([pout[0]]) } else { None } });
        }
        Some(PrecompileCyclesWitness::Keccak256(v))
    } else {
        None // DefaultPrecompilesProcessor answers None for any other address: no callback
    }
}
