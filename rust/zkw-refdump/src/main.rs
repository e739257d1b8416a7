//! zkw-refdump — golden fixtures from the REAL reference.
//!
//! UNBUILT in the repository's own image (no Rust toolchain, no network).  Written against the public API the
//! reference crate itself uses (file:line below are /root/reference/src = matter-labs/era-zk_evm @ v1.4.1):
//!   * `VmState::empty_state` / `push_bootloader_context` / `start_frame` vm_state/mod.rs:188-207, helpers.rs:289-316, :225-246
//!   * `VmState::cycle(&mut self, tracer: &mut DT)` with the debug tracer  cycle.rs:257-260; `GenericNoopTracer` utils.rs:51-92
//!   * `VmWitnessTracer` (10 callbacks)                                   witness_trace/mod.rs:11-72
//!   * `SimpleMemory`, `SimpleDecommitter<true>`, `InMemoryEventSink`     reference_impls/
//!   * `InMemoryStorage`, `DefaultPrecompilesProcessor<true>`             testing/storage.rs, testing/mod.rs:12-40
//!   * `OPCODES_TABLE`, `OPCODES_PRICES`, the variant accessors           cycle.rs:142-184, 341 (zkevm_opcode_defs)
//!
//! Two sub-commands:
//!   dump-isa <out>          the real ISA table + constants in the layout of `zkw_isa_table` (include/zkw.h)
//!   run <inputs> <out>      replays a workload written by tests/golden/make_ref_inputs.py (tapes already encoded
//!                           for the real table) through `VmState::cycle` with a recording tracer and writes every
//!                           instance's trace in the layouts of include/zkw.h (`zkw_cycle_record`, `zkw_mem_query`,
//!                           `zkw_log_query`, `zkw_aux_event`, per-cycle offsets, final `zkw_vm_local_state`)
//! Container format of every file: magic "ZKWREF01", then sections { u32 name_len, name, u64 data_len, data }.
//! tests/test_reference_fixtures.py reads the same format.
use std::collections::HashMap;
use std::io::{Read, Write};

use zk_evm::aux_structures::{DecommittmentQuery, LogQuery, MemoryPage, MemoryQuery, Timestamp};
use zk_evm::block_properties::BlockProperties;
use zk_evm::ethereum_types::{Address, U256};
use zk_evm::reference_impls::{decommitter::SimpleDecommitter, event_sink::InMemoryEventSink, memory::SimpleMemory};
use zk_evm::testing::storage::InMemoryStorage;
use zk_evm::vm_state::{CallStackEntry, PrimitiveValue, VmLocalState, VmState};
use zk_evm::witness_trace::VmWitnessTracer;
use zk_evm::zk_evm_abstractions::precompiles::DefaultPrecompilesProcessor;
use zk_evm::zk_evm_abstractions::vm::{MemoryType, PrecompileCyclesWitness, RefundType};
use zk_evm::zkevm_opcode_defs as defs;
use zk_evm::zkevm_opcode_defs::decoding::encoding_mode_production::EncodingModeProduction;
use zk_evm::zkevm_opcode_defs::decoding::VmEncodingMode;
use zk_evm::GenericNoopTracer; // utils.rs:51-60, re-exported at the crate root (lib.rs:12 `pub use self::utils::*`)

type E = EncodingModeProduction;

// ---------------------------------------------------------------------------------------------------------------
// container
// ---------------------------------------------------------------------------------------------------------------
const MAGIC: &[u8; 8] = b"ZKWREF01";

fn read_container(path: &str) -> anyhow::Result<HashMap<String, Vec<u8>>> {
    let mut buf = vec![];
    std::fs::File::open(path)?.read_to_end(&mut buf)?;
    anyhow::ensure!(&buf[..8] == MAGIC, "bad magic");
    let mut at = 8usize;
    let mut out = HashMap::new();
    while at < buf.len() {
        let nl = u32::from_le_bytes(buf[at..at + 4].try_into()?) as usize;
        at += 4;
        let name = String::from_utf8(buf[at..at + nl].to_vec())?;
        at += nl;
        let dl = u64::from_le_bytes(buf[at..at + 8].try_into()?) as usize;
        at += 8;
        out.insert(name, buf[at..at + dl].to_vec());
        at += dl;
    }
    Ok(out)
}

struct Writer(Vec<u8>);
impl Writer {
    fn new() -> Self {
        Writer(MAGIC.to_vec())
    }
    fn section(&mut self, name: &str, data: &[u8]) {
        self.0.extend_from_slice(&(name.len() as u32).to_le_bytes());
        self.0.extend_from_slice(name.as_bytes());
        self.0.extend_from_slice(&(data.len() as u64).to_le_bytes());
        self.0.extend_from_slice(data);
    }
    fn save(self, path: &str) -> anyhow::Result<()> {
        std::fs::File::create(path)?.write_all(&self.0)?;
        Ok(())
    }
}

fn u256_le(v: &U256) -> [u8; 32] {
    let mut b = [0u8; 32];
    v.to_little_endian(&mut b);
    b
}
fn u256_from_le(b: &[u8]) -> U256 {
    U256::from_little_endian(&b[..32])
}
/// include/zkw.h convention: addresses are the little-endian bytes of the 160-bit integer (H160 is big-endian)
fn address_le(a: &Address) -> [u8; 20] {
    let mut b = a.to_fixed_bytes();
    b.reverse();
    b
}
/// the low 32 bits of an address (H160 is big-endian)
fn address_low_u32(a: &Address) -> u32 {
    let b = a.as_fixed_bytes();
    u32::from_be_bytes([b[16], b[17], b[18], b[19]])
}
fn address_from_le(b: &[u8]) -> Address {
    let mut x = [0u8; 20];
    x.copy_from_slice(&b[..20]);
    x.reverse();
    Address::from(x)
}

// ---------------------------------------------------------------------------------------------------------------
// dump-isa: OPCODES_TABLE / OPCODES_PRICES -> zkw_isa_table (2048 x zkw_isa_entry{opcode, variant, src0_mode, dst0_mode,
// flags, props, reserved u16, price u32} + zkw_isa_consts, 120 B)
// ---------------------------------------------------------------------------------------------------------------
// numbering of include/zkw.h
const OP_INVALID: u8 = 0;
const OPS: [&str; 16] = ["Invalid", "Nop", "Add", "Sub", "Mul", "Div", "Jump", "Context", "Shift", "Binop", "Ptr", "NearCall", "Log", "FarCall", "Ret", "UMA"];
const MODE_REG: u8 = 0;
const PROP_EXPLICIT_PANIC: u8 = 1;
const PROP_KERNEL_ONLY: u8 = 2;
const PROP_STATIC_OK: u8 = 4;
const PROP_SWAP: u8 = 8;
const PROP_SRC0_PTR_OK: u8 = 16;
const PROP_SRC1_PTR_OK: u8 = 32;

/// `Opcode::Add(AddOpcode::Add)` prints as "Add(Add)": outer name = opcode family, inner = variant.  The inner variant's
/// position is taken from the family's `ALL_VARIANTS`-style ordering as printed by the crate (`variant_index` below
/// parses the discriminant through the Debug name table of include/zkw.h).
fn family_and_variant(op: &defs::Opcode) -> (u8, u8) {
    let s = format!("{:?}", op);
    let (fam, inner) = match s.find('(') {
        Some(i) => (&s[..i], s[i + 1..s.len() - 1].to_string()),
        None => (&s[..], String::new()),
    };
    let fam_idx = OPS.iter().position(|n| *n == fam).unwrap_or(OP_INVALID as usize) as u8;
    // variant names in the order of include/zkw.h (ZKW_CTX_*, ZKW_SHIFT_*, ZKW_BINOP_*, ZKW_PTR_*, ZKW_LOG_*, ZKW_FAR_*,
    // ZKW_RET_*, ZKW_UMA_*)
    let table: &[&str] = match fam {
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
    let var_idx = table.iter().position(|n| *n == inner).unwrap_or(0) as u8;
    (fam_idx, var_idx)
}

fn operand_mode(o: &defs::Operand) -> u8 {
    use defs::{ImmMemHandlerFlags as F, Operand, RegOrImmFlags};
    // ZKW_MODE_*: REG 0, STACK_PP 1, STACK_OFF 2, STACK_ABS 3, IMM 4, CODE 5 (mem_ops.rs:37-122, cycle.rs:327-337)
    match o {
        Operand::RegOnly => MODE_REG,
        Operand::RegOrImm(RegOrImmFlags::UseRegOnly) => MODE_REG,
        Operand::RegOrImm(RegOrImmFlags::UseImm16Only) => 4,
        Operand::Full(F::UseRegOnly) => MODE_REG,
        Operand::Full(F::UseStackWithPushPop) => 1,
        Operand::Full(F::UseStackWithOffset) => 2,
        Operand::Full(F::UseAbsoluteOnStack) => 3,
        Operand::Full(F::UseImm16Only) => 4,
        Operand::Full(F::UseCodePage) => 5,
    }
}

fn dump_isa(out: &str) -> anyhow::Result<()> {
    let mut table = vec![0u8; 12 * 2048 + 120];
    for idx in 0..2048usize {
        let v = &defs::OPCODES_TABLE[idx];
        let (fam, var) = family_and_variant(&v.opcode);
        let mut flags = 0u8;
        for (i, f) in v.flags.iter().enumerate() {
            if *f {
                flags |= 1 << i;
            }
        }
        let mut props = 0u8;
        if v.is_explicit_panic() {
            props |= PROP_EXPLICIT_PANIC;
        }
        if v.requires_kernel_mode() {
            props |= PROP_KERNEL_ONLY;
        }
        if v.can_be_used_in_static_context() {
            props |= PROP_STATIC_OK;
        }
        if v.swap_operands() {
            props |= PROP_SWAP;
        }
        if v.opcode.src0_can_be_pointer() {
            props |= PROP_SRC0_PTR_OK;
        }
        if v.opcode.src1_can_be_pointer() {
            props |= PROP_SRC1_PTR_OK;
        }
        let e = &mut table[12 * idx..12 * idx + 12];
        e[0] = fam;
        e[1] = var;
        e[2] = operand_mode(&v.src0_operand_type);
        e[3] = operand_mode(&v.dst0_operand_type);
        e[4] = flags;
        e[5] = props;
        e[8..12].copy_from_slice(&(defs::OPCODES_PRICES[idx] as u32).to_le_bytes());
    }
    // zkw_isa_consts (include/zkw.h; offsets as in era-zk_evm_amd/capi.py ISA_CONSTS)
    let c = &mut table[12 * 2048..];
    let nop = E::nop_encoding();
    let revert = E::exception_revert_encoding();
    c[0..8].copy_from_slice(&nop.to_le_bytes());
    c[8..16].copy_from_slice(&revert.to_le_bytes());
    c[16..20].copy_from_slice(&((revert & 0x7ff) as u32).to_le_bytes());
    c[20..24].copy_from_slice(&((nop & 0x7ff) as u32).to_le_bytes());
    // clip_mode: 0 = saturate, 1 = truncate — decided by the crate's own from_u64_clipped
    let clipped = <<E as VmEncodingMode<8>>::PcOrImm as defs::decoding::AllowedPcOrImm>::from_u64_clipped(0x1_0001).as_u64();
    c[24..28].copy_from_slice(&(if clipped == 0xffff { 0u32 } else { 1u32 }).to_le_bytes());
    let put = |c: &mut [u8], off: usize, v: u32| c[off..off + 4].copy_from_slice(&v.to_le_bytes());
    put(c, 28, defs::TIME_DELTA_PER_CYCLE);
    put(c, 32, defs::NEW_MEMORY_PAGES_PER_FAR_CALL);
    put(c, 36, defs::system_params::VM_MAX_STACK_DEPTH);
    put(c, 40, defs::INITIAL_SP_ON_FAR_CALL as u32);
    put(c, 44, defs::system_params::NEW_FRAME_MEMORY_STIPEND);
    put(c, 48, defs::system_params::MEMORY_GROWTH_ERGS_PER_BYTE);
    put(c, 52, defs::ERGS_PER_CODE_WORD_DECOMMITTMENT);
    put(c, 56, defs::system_params::INITIAL_STORAGE_WRITE_PUBDATA_BYTES as u32);
    put(c, 60, defs::system_params::L1_MESSAGE_PUBDATA_BYTES);
    put(c, 64, defs::uma::MAX_OFFSET_TO_DEREF.low_u32()); // the U256 bound of uma.rs:127
    put(c, 68, address_low_u32(&defs::system_params::DEPLOYER_SYSTEM_CONTRACT_ADDRESS)); // far_call.rs:6,136
    put(c, 72, address_low_u32(&defs::system_params::KECCAK256_ROUND_FUNCTION_PRECOMPILE_FORMAL_ADDRESS) & 0xffff); // testing/tests/precompiles/keccak256.rs:114
    put(c, 76, defs::system_params::SHA256_ROUND_FUNCTION_PRECOMPILE_ADDRESS as u32);
    put(c, 80, defs::system_params::ECRECOVER_INNER_FUNCTION_PRECOMPILE_ADDRESS as u32);
    c[84] = defs::system_params::STORAGE_AUX_BYTE;
    c[85] = defs::system_params::EVENT_AUX_BYTE;
    c[86] = defs::system_params::L1_MESSAGE_AUX_BYTE;
    c[87] = defs::system_params::PRECOMPILE_AUX_BYTE;
    put(c, 92, defs::BOOTLOADER_CALLDATA_PAGE); // memory.rs:11 imports it from the crate root
    put(c, 88, 0); // ecrecover_input_layout: (hash, v, r, s) vs (hash, r, s, v) is settled by the ecrecover fixture itself
    let mut w = Writer::new();
    w.section("isa", &table);
    w.save(out)
}

// ---------------------------------------------------------------------------------------------------------------
// run: the recording tracer
// ---------------------------------------------------------------------------------------------------------------
#[derive(Clone, Debug, Default)]
struct Recorder {
    seq: u32,
    records: Vec<u8>,  // n x 512 B zkw_cycle_record
    mem: Vec<u8>,      // n x 48 B
    log: Vec<u8>,      // n x 128 B
    aux: Vec<u8>,      // n x 256 B
    mem_off: Vec<u32>,
    log_off: Vec<u32>,
    aux_off: Vec<u32>,
    cold: (u32, u32, u16, u32, u128),
}

fn entry_c(e: &CallStackEntry<8, E>) -> [u8; 112] {
    // zkw_callstack_entry (include/zkw.h; offsets as in capi.py CALLSTACK_ENTRY)
    let mut b = [0u8; 112];
    b[0..20].copy_from_slice(&address_le(&e.this_address));
    b[20..40].copy_from_slice(&address_le(&e.msg_sender));
    b[40..60].copy_from_slice(&address_le(&e.code_address));
    b[60..64].copy_from_slice(&e.base_memory_page.0.to_le_bytes());
    b[64..68].copy_from_slice(&e.code_page.0.to_le_bytes());
    b[68..70].copy_from_slice(&e.sp.to_le_bytes());
    b[70..72].copy_from_slice(&e.pc.to_le_bytes());
    b[72..74].copy_from_slice(&e.exception_handler_location.to_le_bytes());
    b[74] = e.is_static as u8;
    b[75] = e.is_local_frame as u8;
    b[76..80].copy_from_slice(&e.ergs_remaining.to_le_bytes());
    b[80] = e.this_shard_id;
    b[81] = e.caller_shard_id;
    b[82] = e.code_shard_id;
    b[88..104].copy_from_slice(&e.context_u128_value.to_le_bytes());
    b[104..108].copy_from_slice(&e.heap_bound.to_le_bytes());
    b[108..112].copy_from_slice(&e.aux_heap_bound.to_le_bytes());
    b
}

fn state_c(s: &VmLocalState<8, E>) -> [u8; 680] {
    // zkw_vm_local_state (capi.py VM_LOCAL_STATE)
    let mut b = [0u8; 680];
    b[0..32].copy_from_slice(&u256_le(&s.previous_code_word));
    let mut bm = 0u16;
    for (i, r) in s.registers.iter().enumerate() {
        b[32 + 32 * i..64 + 32 * i].copy_from_slice(&u256_le(&r.value));
        if r.is_pointer {
            bm |= 1 << i;
        }
    }
    b[512..514].copy_from_slice(&bm.to_le_bytes());
    b[514] = (s.flags.overflow_or_less_than_flag as u8) | ((s.flags.equality_flag as u8) << 1) | ((s.flags.greater_than_flag as u8) << 2);
    b[515] = s.pending_exception as u8;
    b[516..520].copy_from_slice(&s.previous_code_memory_page.0.to_le_bytes());
    b[520..524].copy_from_slice(&s.timestamp.to_le_bytes());
    b[524..528].copy_from_slice(&s.monotonic_cycle_counter.to_le_bytes());
    b[528..532].copy_from_slice(&s.spent_pubdata_counter.to_le_bytes());
    b[532..536].copy_from_slice(&s.memory_page_counter.to_le_bytes());
    b[536..540].copy_from_slice(&s.absolute_execution_step.to_le_bytes());
    b[540..544].copy_from_slice(&s.current_ergs_per_pubdata_byte.to_le_bytes());
    b[544..546].copy_from_slice(&s.tx_number_in_block.to_le_bytes());
    b[546..548].copy_from_slice(&s.previous_super_pc.to_le_bytes());
    b[548..552].copy_from_slice(&(s.callstack.inner.len() as u32).to_le_bytes());
    b[552..568].copy_from_slice(&s.context_u128_register.to_le_bytes());
    b[568..680].copy_from_slice(&entry_c(&s.callstack.current));
    b
}

impl Recorder {
    fn next_seq(&mut self) -> u8 {
        let s = self.seq.min(255) as u8;
        self.seq += 1;
        s
    }
    fn push_mem(&mut self, q: &MemoryQuery, kind: u8) {
        // zkw_mem_query (48 B): timestamp, page, index, lane, seq, meta, reserved, value
        let mut b = [0u8; 48];
        b[0..4].copy_from_slice(&q.timestamp.0.to_le_bytes());
        b[4..8].copy_from_slice(&q.location.page.0.to_le_bytes());
        b[8..12].copy_from_slice(&q.location.index.0.to_le_bytes());
        b[13] = self.next_seq();
        let ty = match q.location.memory_type {
            MemoryType::Stack => 0u8,
            MemoryType::Code => 1,
            MemoryType::Heap => 2,
            MemoryType::AuxHeap => 3,
            MemoryType::FatPointer => 4,
        };
        b[14] = ty | ((q.value_is_pointer as u8) << 3) | ((q.rw_flag as u8) << 4) | (kind << 5);
        b[16..48].copy_from_slice(&u256_le(&q.value));
        self.mem.extend_from_slice(&b);
    }
    fn push_log(&mut self, q: &LogQuery, kind: u8) {
        // zkw_log_query (128 B)
        let mut b = [0u8; 128];
        b[0..32].copy_from_slice(&u256_le(&q.key));
        b[32..64].copy_from_slice(&u256_le(&q.read_value));
        b[64..96].copy_from_slice(&u256_le(&q.written_value));
        b[96..116].copy_from_slice(&address_le(&q.address));
        b[116..120].copy_from_slice(&q.timestamp.0.to_le_bytes());
        b[120..122].copy_from_slice(&q.tx_number_in_block.to_le_bytes());
        b[122] = q.aux_byte;
        b[123] = q.shard_id;
        b[124] = (q.rw_flag as u8) | ((q.rollback as u8) << 1) | ((q.is_service as u8) << 2);
        b[125] = kind;
        b[127] = self.next_seq();
        self.log.extend_from_slice(&b);
    }
    fn push_aux(&mut self, ty: u8, flag: u8, a: u32, bb: u32, c: u32, payload: &[u8]) {
        // zkw_aux_event (256 B): type, lane, seq, flag, a, b, c, raw[240]
        let mut b = [0u8; 256];
        b[0] = ty;
        b[2] = self.next_seq();
        b[3] = flag;
        b[4..8].copy_from_slice(&a.to_le_bytes());
        b[8..12].copy_from_slice(&bb.to_le_bytes());
        b[12..16].copy_from_slice(&c.to_le_bytes());
        b[16..16 + payload.len()].copy_from_slice(payload);
        self.aux.extend_from_slice(&b);
    }
}

impl VmWitnessTracer<8, E> for Recorder {
    fn start_new_execution_cycle(&mut self, _s: &VmLocalState<8, E>) {
        self.seq = 0;
    }
    fn end_execution_cycle(&mut self, s: &VmLocalState<8, E>) {
        // cold fields: one ZKW_AUX_COLD_STATE (4) event in the cycle that changed them
        let cold = (s.spent_pubdata_counter, s.current_ergs_per_pubdata_byte, s.tx_number_in_block, s.memory_page_counter, s.context_u128_register);
        if cold != self.cold {
            let mut p = [0u8; 20];
            p[0..16].copy_from_slice(&cold.4.to_le_bytes());
            p[16..20].copy_from_slice(&cold.3.to_le_bytes());
            self.push_aux(4, 0, cold.0, cold.1, cold.2 as u32, &p);
            self.cold = cold;
        }
        // zkw_cycle_record: 15 registers + zkw_cycle_tail (capi.py CYCLE_TAIL)
        let mut r = [0u8; 512];
        let mut bm = 0u16;
        for (i, v) in s.registers.iter().enumerate() {
            r[32 * i..32 * i + 32].copy_from_slice(&u256_le(&v.value));
            if v.is_pointer {
                bm |= 1 << i;
            }
        }
        let cur = &s.callstack.current;
        let t = &mut r[480..512];
        t[0..2].copy_from_slice(&bm.to_le_bytes());
        t[2] = (s.flags.overflow_or_less_than_flag as u8) | ((s.flags.equality_flag as u8) << 1) | ((s.flags.greater_than_flag as u8) << 2) | ((s.pending_exception as u8) << 3);
        t[4..6].copy_from_slice(&cur.pc.to_le_bytes());
        t[6..8].copy_from_slice(&cur.sp.to_le_bytes());
        t[8..12].copy_from_slice(&cur.ergs_remaining.to_le_bytes());
        t[12..16].copy_from_slice(&s.timestamp.to_le_bytes());
        t[16..20].copy_from_slice(&cur.heap_bound.to_le_bytes());
        t[20..24].copy_from_slice(&cur.aux_heap_bound.to_le_bytes());
        t[24..26].copy_from_slice(&(s.callstack.inner.len() as u16).to_le_bytes());
        t[26..28].copy_from_slice(&s.previous_super_pc.to_le_bytes());
        let n = self.mem_off.len();
        let counts = |now: usize, offs: &Vec<u32>| (now as u32 - offs[n - 1]).min(255);
        let cnt = counts(self.mem.len() / 48, &self.mem_off) | (counts(self.log.len() / 128, &self.log_off) << 8) | (counts(self.aux.len() / 256, &self.aux_off) << 16);
        t[28..32].copy_from_slice(&cnt.to_le_bytes());
        self.records.extend_from_slice(&r);
        self.mem_off.push((self.mem.len() / 48) as u32);
        self.log_off.push((self.log.len() / 128) as u32);
        self.aux_off.push((self.aux.len() / 256) as u32);
    }
    fn add_memory_query(&mut self, _cc: u32, q: MemoryQuery) {
        self.push_mem(&q, 0);
    }
    fn record_refund_for_query(&mut self, _cc: u32, q: LogQuery, _refund: RefundType) {
        self.push_log(&q, 1);
    }
    fn add_log_query(&mut self, _cc: u32, q: LogQuery) {
        self.push_log(&q, 0);
    }
    fn add_decommittment(&mut self, _cc: u32, q: DecommittmentQuery, _mem_witness: Vec<U256>) {
        // ZKW_AUX_DECOMMIT (3): flag = is_fresh, a = timestamp, b = memory_page, c = decommitted_length (the blob id of
        // the device's bookkeeping is not a reference notion: the comparison masks the upper half of `c`)
        self.push_aux(3, q.is_fresh as u8, q.timestamp.0, q.memory_page.0, q.decommitted_length as u32, &u256_le(&q.hash));
    }
    fn add_precompile_call_result(&mut self, _cc: u32, _call: LogQuery, mem_in: Vec<MemoryQuery>, mem_out: Vec<MemoryQuery>, _rounds: PrecompileCyclesWitness) {
        for q in mem_in.iter() {
            self.push_mem(q, 1);
        }
        for q in mem_out.iter() {
            self.push_mem(q, 2);
        }
    }
    fn start_new_execution_context(&mut self, _cc: u32, prev: &CallStackEntry<8, E>, next: &CallStackEntry<8, E>) {
        let mut p = [0u8; 224];
        p[0..112].copy_from_slice(&entry_c(prev));
        p[112..224].copy_from_slice(&entry_c(next));
        self.push_aux(1, (!next.is_local_frame) as u8, 0, 0, 0, &p);
    }
    fn finish_execution_context(&mut self, _cc: u32, panicked: bool) {
        self.push_aux(2, panicked as u8, 0, 0, 0, &[]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// run
// ---------------------------------------------------------------------------------------------------------------
fn entry_from_c(b: &[u8]) -> CallStackEntry<8, E> {
    let u16at = |o: usize| u16::from_le_bytes([b[o], b[o + 1]]);
    let u32at = |o: usize| u32::from_le_bytes(b[o..o + 4].try_into().unwrap());
    CallStackEntry {
        this_address: address_from_le(&b[0..20]),
        msg_sender: address_from_le(&b[20..40]),
        code_address: address_from_le(&b[40..60]),
        base_memory_page: MemoryPage(u32at(60)),
        code_page: MemoryPage(u32at(64)),
        sp: u16at(68),
        pc: u16at(70),
        exception_handler_location: u16at(72),
        is_static: b[74] != 0,
        is_local_frame: b[75] != 0,
        ergs_remaining: u32at(76),
        this_shard_id: b[80],
        caller_shard_id: b[81],
        code_shard_id: b[82],
        context_u128_value: u128::from_le_bytes(b[88..104].try_into().unwrap()),
        heap_bound: u32at(104),
        aux_heap_bound: u32at(108),
    }
}

fn words(b: &[u8]) -> Vec<U256> {
    b.chunks(32).map(u256_from_le).collect()
}

fn run(inputs: &str, out: &str) -> anyhow::Result<()> {
    let c = read_container(inputs)?;
    let u32s = |name: &str| -> Vec<u32> { c[name].chunks(4).map(|x| u32::from_le_bytes(x.try_into().unwrap())).collect() };
    let meta = u32s("meta"); // n_instances, n_cycles, inner_depth, n_blobs, heap_words, zkporter_is_available
    let (n, n_cycles, depth, n_blobs) = (meta[0] as usize, meta[1] as usize, meta[2] as usize, meta[3] as usize);
    let blobs: Vec<Vec<U256>> = (0..n_blobs).map(|i| words(&c[&format!("blob{}", i)])).collect();
    let preimages = &c["preimages"]; // (hash 32 B, blob u32) x k
    let code_pages = u32s("code_pages"); // (first, count, page, blob) x k
    let default_aa = u256_from_le(&c["default_aa_code_hash"]);
    let mut w = Writer::new();
    w.section("meta", &c["meta"]);
    for i in 0..n {
        let st = &c["states"][680 * i..680 * (i + 1)];
        let inner = &c["inner"][112 * depth * i..112 * depth * (i + 1)];
        // (the type annotation fixes the hasher parameter: only `SimpleMemory<RandomState>` implements `Memory`, memory.rs:403)
        let memory: SimpleMemory = SimpleMemory::new_without_preallocations();
        let mut storage = InMemoryStorage::new();
        let mut decommitter = SimpleDecommitter::<true>::new();
        decommitter.populate(
            preimages.chunks(36).map(|p| (u256_from_le(&p[0..32]), blobs[u32::from_le_bytes(p[32..36].try_into().unwrap()) as usize].clone())).collect(),
        );
        if let Some(s) = c.get(&format!("storage{}", i)) {
            // zkw_storage_slot (88 B): key, value, address[20], shard_id
            storage.populate(s.chunks(88).map(|e| (e[84], address_from_le(&e[64..84]), u256_from_le(&e[0..32]), u256_from_le(&e[32..64]))).collect());
        }
        let mut vm = VmState::<_, _, _, _, _, _, 8, E>::empty_state(
            storage,
            memory,
            InMemoryEventSink::new(),
            DefaultPrecompilesProcessor::<true>,
            decommitter,
            Recorder::default(),
            BlockProperties { default_aa_code_hash: default_aa, zkporter_is_available: meta[5] != 0 },
        );
        // the frames alive at the start: what the host did before handing over (helpers.rs:289-316)
        let frames: Vec<CallStackEntry<8, E>> = (1..depth).map(|d| entry_from_c(&inner[112 * d..112 * (d + 1)])).chain(std::iter::once(entry_from_c(&st[568..680]))).collect();
        for f in frames.iter() {
            // The exact entries (ergs included) are written over the callstack below; here the frames only have to exist
            // in the oracles.  `push_bootloader_context` subtracts the new frame's ergs from the current one and asserts
            // that it does not underflow (helpers.rs:295-302), so the frame is pushed with no ergs; a far frame also
            // starts a global memory frame (helpers.rs:308-315), a near-call frame is `start_frame` alone (:225-246).
            let pushed = CallStackEntry { ergs_remaining: 0, ..*f };
            if f.is_local_frame {
                vm.start_frame(0, pushed);
            } else {
                vm.push_bootloader_context(0, pushed);
            }
        }
        // code pages, heap image of the first far frame
        let mut pages = vec![];
        for k in code_pages.chunks(4) {
            if (k[0] as usize..(k[0] + k[1]) as usize).contains(&i) {
                pages.push((k[2], blobs[k[3] as usize].clone()));
            }
        }
        vm.memory.populate_code(pages);
        if let Some(h) = c.get(&format!("heap{}", i)) {
            vm.memory.populate_heap(words(h));
        }
        // the scalar state of zkw_vm_local_state on top of the pushed frames
        {
            let u32at = |o: usize| u32::from_le_bytes(st[o..o + 4].try_into().unwrap());
            let ls = &mut vm.local_state;
            ls.previous_code_word = u256_from_le(&st[0..32]);
            let bm = u16::from_le_bytes([st[512], st[513]]);
            for r in 0..15 {
                ls.registers[r] = PrimitiveValue { value: u256_from_le(&st[32 + 32 * r..64 + 32 * r]), is_pointer: (bm >> r) & 1 != 0 };
            }
            ls.flags.overflow_or_less_than_flag = st[514] & 1 != 0;
            ls.flags.equality_flag = st[514] & 2 != 0;
            ls.flags.greater_than_flag = st[514] & 4 != 0;
            ls.pending_exception = st[515] != 0;
            ls.previous_code_memory_page = MemoryPage(u32at(516));
            ls.timestamp = u32at(520);
            ls.monotonic_cycle_counter = u32at(524);
            ls.spent_pubdata_counter = u32at(528);
            ls.memory_page_counter = u32at(532);
            ls.absolute_execution_step = u32at(536);
            ls.current_ergs_per_pubdata_byte = u32at(540);
            ls.tx_number_in_block = u16::from_le_bytes([st[544], st[545]]);
            ls.previous_super_pc = u16::from_le_bytes([st[546], st[547]]);
            ls.context_u128_register = u128::from_le_bytes(st[552..568].try_into().unwrap());
            // push_bootloader_context derives ergs; the inputs carry the exact entries
            ls.callstack.current = entry_from_c(&st[568..680]);
            for d in 0..depth {
                ls.callstack.inner[d] = entry_from_c(&inner[112 * d..112 * (d + 1)]);
            }
        }
        vm.witness_tracer = Recorder::default();
        vm.witness_tracer.mem_off.push(0);
        vm.witness_tracer.log_off.push(0);
        vm.witness_tracer.aux_off.push(0);
        vm.witness_tracer.cold = (
            vm.local_state.spent_pubdata_counter,
            vm.local_state.current_ergs_per_pubdata_byte,
            vm.local_state.tx_number_in_block,
            vm.local_state.memory_page_counter,
            vm.local_state.context_u128_register,
        );
        // the debug tracer argument of `cycle` (cycle.rs:257-260): the crate's own no-op implementation, whose four
        // CALL_* constants are false (tracing.rs:43-46) — the hooks are compiled out
        let mut debug_tracer = GenericNoopTracer::<SimpleMemory>::new();
        let mut status = 0u32; // ZKW_STATUS_RUNNING
        for _ in 0..n_cycles {
            if vm.execution_has_ended() {
                status = 1; // ZKW_STATUS_ENDED
                break;
            }
            let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| vm.cycle(&mut debug_tracer)));
            match r {
                Ok(Ok(())) => {}
                Ok(Err(_)) => {
                    status = 2; // ZKW_STATUS_UNKNOWN_CODE_HASH (decommitter.rs:54-56: the only Err source)
                    break;
                }
                Err(_) => {
                    status = 3; // ZKW_STATUS_REFERENCE_PANIC
                    break;
                }
            }
        }
        if status == 0 && vm.execution_has_ended() {
            status = 1;
        }
        let t = &vm.witness_tracer;
        // a cycle that did not complete leaves partial records: keep whole cycles only (as the product and the oracle do)
        let done = t.mem_off.len() - 1;
        w.section(&format!("status{}", i), &status.to_le_bytes());
        w.section(&format!("rec{}", i), &t.records[..512 * done]);
        w.section(&format!("mem{}", i), &t.mem[..48 * t.mem_off[done] as usize]);
        w.section(&format!("log{}", i), &t.log[..128 * t.log_off[done] as usize]);
        w.section(&format!("aux{}", i), &t.aux[..256 * t.aux_off[done] as usize]);
        let offs = |v: &Vec<u32>| v.iter().flat_map(|x| x.to_le_bytes()).collect::<Vec<u8>>();
        w.section(&format!("memoff{}", i), &offs(&t.mem_off));
        w.section(&format!("logoff{}", i), &offs(&t.log_off));
        w.section(&format!("auxoff{}", i), &offs(&t.aux_off));
        w.section(&format!("final{}", i), &state_c(&vm.local_state));
    }
    w.save(out)
}

fn main() -> anyhow::Result<()> {
    let a: Vec<String> = std::env::args().collect();
    match a.get(1).map(|s| s.as_str()) {
        Some("dump-isa") if a.len() == 3 => dump_isa(&a[2]),
        Some("run") if a.len() == 4 => run(&a[2], &a[3]),
        _ => anyhow::bail!("usage: zkw-refdump dump-isa <out.bin> | run <inputs.bin> <out.bin>"),
    }
}
