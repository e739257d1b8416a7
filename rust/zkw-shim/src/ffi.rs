//! `#[repr(C)]` mirror of include/zkw.h — field for field (sizes are asserted against `zkw_abi_sizeof` in `check_abi`).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy, Default, Debug, PartialEq, Eq)]
pub struct zkw_u256 {
    pub l: [u64; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_isa_entry {
    pub opcode: u8,
    pub variant: u8,
    pub src0_mode: u8,
    pub dst0_mode: u8,
    pub flags: u8,
    pub props: u8,
    pub reserved: u16,
    pub price: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_isa_consts {
    pub nop_encoding: u64,
    pub exception_revert_encoding: u64,
    pub panic_variant_idx: u32,
    pub nop_variant_idx: u32,
    pub clip_mode: u32,
    pub time_delta_per_cycle: u32,
    pub new_memory_pages_per_far_call: u32,
    pub vm_max_stack_depth: u32,
    pub initial_sp_on_far_call: u32,
    pub new_frame_memory_stipend: u32,
    pub memory_growth_ergs_per_byte: u32,
    pub ergs_per_code_word_decommittment: u32,
    pub initial_storage_write_pubdata_bytes: u32,
    pub l1_message_pubdata_bytes: u32,
    pub max_offset_to_deref_low: u32,
    pub deployer_address_low: u32,
    pub keccak_precompile_address: u32,
    pub sha256_precompile_address: u32,
    pub ecrecover_precompile_address: u32,
    pub storage_aux_byte: u8,
    pub event_aux_byte: u8,
    pub l1_message_aux_byte: u8,
    pub precompile_aux_byte: u8,
    pub ecrecover_input_layout: u32,
    pub bootloader_calldata_page: u32,
    /// far_call.rs:506-508,573-610: byte 0 CALL_IMPLICIT_CALLDATA_FAT_PTR_REGISTER, 1 CALL_IMPLICIT_CONSTRUCTOR_MARKER_REGISTER, 2 CALL_IMPLICIT_PARAMETER_REG_IDX
    pub call_regs: u32,
    /// bytes 0 / 1: CALL_SYSTEM_ABI_REGISTERS first / end, bytes 2 / 3: CALL_RESERVED_RANGE first / end (ends exclusive)
    pub call_ranges: u32,
    /// ret.rs:213-233: byte 0 RET_IMPLICIT_RETURNDATA_PARAMS_REGISTER, bytes 1..3 RET_RESERVED_REGISTER_0..2
    pub ret_regs: u32,
    /// FarCallForwardPageType as the ABI byte: byte 0 UseHeap, 1 ForwardFatPointer, 2 UseAuxHeap
    pub forwarding_codes: u32,
    pub unmapped_page: u32,
    pub reserved0: u32,
    /// ptr::MAX_OFFSET_FOR_ADD_SUB (ptr.rs:47)
    pub max_offset_for_add_sub: u64,
    /// bit 8 * field + (lt_of | eq << 1 | gt << 2): the Condition the 3-bit field names holds (cycle.rs:193-209)
    pub condition_lut: u64,
}

#[repr(C)]
pub struct zkw_isa_table {
    pub entries: [zkw_isa_entry; 2048],
    pub consts: zkw_isa_consts,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_callstack_entry {
    pub this_address: [u8; 20],
    pub msg_sender: [u8; 20],
    pub code_address: [u8; 20],
    pub base_memory_page: u32,
    pub code_page: u32,
    pub sp: u16,
    pub pc: u16,
    pub exception_handler_location: u16,
    pub is_static: u8,
    pub is_local_frame: u8,
    pub ergs_remaining: u32,
    pub this_shard_id: u8,
    pub caller_shard_id: u8,
    pub code_shard_id: u8,
    pub reserved0: u8,
    pub reserved1: u32,
    pub context_u128_value: [u64; 2],
    pub heap_bound: u32,
    pub aux_heap_bound: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_vm_local_state {
    pub previous_code_word: zkw_u256,
    pub registers: [zkw_u256; 15],
    pub register_ptr_bitmap: u16,
    pub flags: u8,
    pub pending_exception: u8,
    pub previous_code_memory_page: u32,
    pub timestamp: u32,
    pub monotonic_cycle_counter: u32,
    pub spent_pubdata_counter: u32,
    pub memory_page_counter: u32,
    pub absolute_execution_step: u32,
    pub current_ergs_per_pubdata_byte: u32,
    pub tx_number_in_block: u16,
    pub previous_super_pc: u16,
    pub callstack_depth: u32,
    pub context_u128_register: [u64; 2],
    pub current: zkw_callstack_entry,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_cycle_tail {
    pub register_ptr_bitmap: u16,
    pub flags: u8, // bits 0..2 lt/eq/gt, bit 3 pending_exception
    pub reserved0: u8,
    pub pc: u16,
    pub sp: u16,
    pub ergs_remaining: u32,
    pub timestamp: u32,
    pub heap_bound: u32,
    pub aux_heap_bound: u32,
    pub callstack_depth: u16,
    pub previous_super_pc: u16,
    pub event_counts: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_cycle_record {
    pub registers: [zkw_u256; 15],
    pub tail: zkw_cycle_tail,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_mem_query {
    pub timestamp: u32,
    pub page: u32,
    pub index: u32,
    pub lane: u8,
    pub seq: u8,
    pub meta: u8, // bits 0-2 MemoryType, 3 value_is_pointer, 4 rw_flag, 5-7 kind (0 plain, 1 precompile read, 2 precompile write)
    pub reserved0: u8,
    pub value: zkw_u256,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_log_query {
    pub key: zkw_u256,
    pub read_value: zkw_u256,
    pub written_value: zkw_u256,
    pub address: [u8; 20],
    pub timestamp: u32,
    pub tx_number_in_block: u16,
    pub aux_byte: u8,
    pub shard_id: u8,
    pub bools: u8, // 1 rw_flag, 2 rollback, 4 is_service
    pub kind: u8,  // 0 add_log_query, 1 record_refund_for_query
    pub lane: u8,
    pub seq: u8,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_aux_frame {
    pub previous: zkw_callstack_entry,
    pub next: zkw_callstack_entry,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_aux_cold {
    pub context_u128_register: [u64; 2],
    pub memory_page_counter: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub union zkw_aux_payload {
    pub frame: zkw_aux_frame,
    pub hash: zkw_u256,
    pub cold: zkw_aux_cold,
    pub raw: [u8; 240],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_aux_event {
    pub r#type: u8, // 1 FRAME_START, 2 FRAME_FINISH, 3 DECOMMIT, 4 COLD_STATE
    pub lane: u8,
    pub seq: u8,
    pub flag: u8,
    pub a: u32,
    pub b: u32,
    pub c: u32,
    pub u: zkw_aux_payload,
}

#[repr(C)]
pub struct zkw_instance_trace {
    pub status: u32,
    pub n_cycles: u32,
    pub n_mem: u32,
    pub n_log: u32,
    pub n_aux: u32,
    pub reserved0: u32,
    pub records: *const zkw_cycle_record,
    pub mem: *const zkw_mem_query,
    pub log: *const zkw_log_query,
    pub aux: *const zkw_aux_event,
    pub mem_off: *const u32,
    pub log_off: *const u32,
    pub aux_off: *const u32,
    pub final_state: zkw_vm_local_state,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct zkw_limits {
    pub max_cycles: u32,
    pub max_far_frames: u32,
    pub max_callstack_depth: u32,
    pub stack_words: u32,
    pub heap_words: u32,
    pub aux_heap_words: u32,
    pub storage_slots: u32,
    pub storage_journal: u32,
    pub max_mem_queries: u32,
    pub max_log_queries: u32,
    pub max_aux_events: u32,
    pub lanes_per_wave: u32,
    pub max_reg_deltas: u32,
    pub reserved: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_storage_slot {
    pub key: zkw_u256,
    pub value: zkw_u256,
    pub address: [u8; 20],
    pub shard_id: u8,
    pub reserved0: [u8; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_block_properties {
    pub default_aa_code_hash: zkw_u256,
    pub zkporter_is_available: u32,
    pub reserved0: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct zkw_run_stats {
    pub cycles: u64,
    pub mem_queries: u64,
    pub log_queries: u64,
    pub aux_events: u64,
    pub instances_ended: u64,
    pub instances_failed: u64,
    pub kernel_ms: f64,
    pub reg_deltas: u64,
}

/// EventMessage, reference_impls/event_sink.rs:7-14
#[repr(C)]
#[derive(Clone, Copy)]
pub struct zkw_event_message {
    pub shard_id: u8,
    pub is_first: u8,
    pub tx_number_in_block: u16,
    pub address: [u8; 20],
    pub key: zkw_u256,
    pub value: zkw_u256,
}

/// get_final_net_states (testing/mod.rs:42-71) of one instance: library-owned arrays
#[repr(C)]
pub struct zkw_net_state {
    pub n_storage_history: u32,
    pub n_event_history: u32,
    pub n_events: u32,
    pub n_l1_messages: u32,
    pub n_final_storage: u32,
    pub reserved0: u32,
    pub storage_history: *const zkw_log_query,
    pub event_history: *const zkw_log_query,
    pub events: *const zkw_event_message,
    pub l1_messages: *const zkw_event_message,
    pub final_storage: *const zkw_storage_slot,
}

pub enum zkw_ctx {}
pub enum zkw_batch {}
pub enum zkw_comm {}
pub enum zkw_delivery {}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct zkw_delivered {
    pub bytes: u64,
    pub pack_ms: f64,
    pub n_batches: u32,
    pub n_waves: u32,
    pub overflow: u32,
    pub link_flags: u32,
}
pub type zkw_cycle_fn = unsafe extern "C" fn(user: *mut c_void, thread: u32, batch_index: u32, instance: u32, cycle: u32, state_after: *const zkw_cycle_record, mem: *const zkw_mem_query,
                                            n_mem: u32, log: *const zkw_log_query, n_log: u32, aux: *const zkw_aux_event, n_aux: u32);
#[repr(C)]
pub struct zkw_comm_id {
    pub bytes: [u8; 128],
}

pub const ZKW_OK: c_int = 0;
pub const ZKW_STATUS_RUNNING: u32 = 0;
pub const ZKW_STATUS_ENDED: u32 = 1;
pub const ZKW_STATUS_UNKNOWN_CODE_HASH: u32 = 2;
pub const ZKW_STATUS_REFERENCE_PANIC: u32 = 3;
pub const ZKW_STATUS_LIMIT: u32 = 4;
/* zkw_ctx_set_option (include/zkw.h): the options a caller of the delivery ring tunes */
pub const ZKW_OPT_PACK_BLOCKS: u32 = 9;
pub const ZKW_OPT_STAGING_BUFFERS: u32 = 10;
pub const ZKW_OPT_READ_VALUES: u32 = 11;
pub const ZKW_OPT_LINK_FLAGS_OFF: u32 = 12;
pub const ZKW_OPT_LINK_SELFCHECK: u32 = 13;

extern "C" {
    pub fn zkw_ctx_create(device: c_int, out: *mut *mut zkw_ctx) -> c_int;
    pub fn zkw_ctx_destroy(ctx: *mut zkw_ctx);
    pub fn zkw_last_error(ctx: *mut zkw_ctx) -> *const c_char;
    pub fn zkw_ctx_set_isa(ctx: *mut zkw_ctx, table: *const zkw_isa_table) -> c_int;
    pub fn zkw_ctx_set_option(ctx: *mut zkw_ctx, option: u32, value: u64) -> c_int;
    pub fn zkw_batch_create(ctx: *mut zkw_ctx, n_instances: u32, limits: *const zkw_limits, out: *mut *mut zkw_batch) -> c_int;
    pub fn zkw_batch_destroy(batch: *mut zkw_batch);
    pub fn zkw_batch_add_code_blob(batch: *mut zkw_batch, words: *const zkw_u256, n_words: u32, blob_id: *mut u32) -> c_int;
    pub fn zkw_batch_add_decommit_preimage(batch: *mut zkw_batch, hash: *const zkw_u256, blob_id: u32) -> c_int;
    pub fn zkw_batch_set_code_page(batch: *mut zkw_batch, first: u32, count: u32, page: u32, blob_id: u32) -> c_int;
    pub fn zkw_batch_set_state(batch: *mut zkw_batch, first: u32, count: u32, states: *const zkw_vm_local_state, inner: *const zkw_callstack_entry, inner_depth: u32) -> c_int;
    pub fn zkw_batch_set_heap(batch: *mut zkw_batch, instance: u32, words: *const zkw_u256, n_words: u32) -> c_int;
    pub fn zkw_batch_set_storage(batch: *mut zkw_batch, instance: u32, slots: *const zkw_storage_slot, n_slots: u32) -> c_int;
    pub fn zkw_batch_set_block_properties(batch: *mut zkw_batch, p: *const zkw_block_properties) -> c_int;
    pub fn zkw_batch_upload(batch: *mut zkw_batch) -> c_int;
    pub fn zkw_batch_reset(batch: *mut zkw_batch, stream: *mut c_void) -> c_int;
    pub fn zkw_batch_run(batch: *mut zkw_batch, max_cycles: u32, stream: *mut c_void) -> c_int;
    pub fn zkw_batches_step(batches: *const *mut zkw_batch, n: u32, max_cycles: u32, queue_mask: u32, stream: *mut c_void) -> c_int;
    pub fn zkw_batches_reset(batches: *const *mut zkw_batch, n: u32, stream: *mut c_void) -> c_int;
    pub fn zkw_batches_step_prepared(batches: *const *mut zkw_batch, n: u32, max_cycles: u32, queue_mask: u32, stream: *mut c_void) -> c_int;
    pub fn zkw_batch_commit(batch: *mut zkw_batch, queue_mask: u32, stream: *mut c_void) -> c_int;
    pub fn zkw_batch_net_states(batch: *mut zkw_batch, stream: *mut c_void) -> c_int;
    pub fn zkw_batch_get_net_state(batch: *mut zkw_batch, instance: u32, out: *mut zkw_net_state) -> c_int;
    pub fn zkw_batch_expand_records(batch: *mut zkw_batch, first: u32, count: u32, dst_device: *mut c_void, instance_stride: u64, cycle_stride: u64, stream: *mut c_void) -> c_int;
    pub fn zkw_batches_expand_records(batches: *const *mut zkw_batch, n: u32, dst_device: *const *mut c_void, instance_stride: u64, cycle_stride: u64, stream: *mut c_void) -> c_int;
    pub fn zkw_batch_sync(batch: *mut zkw_batch) -> c_int;
    pub fn zkw_batch_get_stats(batch: *mut zkw_batch, out: *mut zkw_run_stats) -> c_int;
    pub fn zkw_batch_get_instance_trace(batch: *mut zkw_batch, instance: u32, out: *mut zkw_instance_trace) -> c_int;
    pub fn zkw_batch_get_page(batch: *mut zkw_batch, instance: u32, page: u32, first_word: u32, n_words: u32, out: *mut zkw_u256) -> c_int;
    pub fn zkw_batch_set_bootloader_calldata(batch: *mut zkw_batch, instance: u32, words: *const zkw_u256, n_words: u32) -> c_int;
    pub fn zkw_batch_get_commitments(batch: *mut zkw_batch, out: *mut u64) -> c_int;
    pub fn zkw_comm_probe() -> c_int;
    pub fn zkw_comm_get_unique_id(out: *mut zkw_comm_id) -> c_int;
    pub fn zkw_comm_create_rccl(ctx: *mut zkw_ctx, rank: c_int, world: c_int, id: *const zkw_comm_id, out: *mut *mut zkw_comm) -> c_int;
    pub fn zkw_comm_destroy(comm: *mut zkw_comm);
    pub fn zkw_reduce_commitments(comm: *mut zkw_comm, batches: *const *mut zkw_batch, n_batches: u32, queue_mask: u32, gathered: *mut c_void,
                                  n_max_out: *mut u32, sizes_out: *mut u32, total: *mut zkw_run_stats, stream: *mut c_void) -> c_int;
    // delivery to the host: whole steps in a persistent pinned ring (include/zkw.h)
    pub fn zkw_delivery_create(ctx: *mut zkw_ctx, n_slots: u32, slot_bytes: u64, host_threads: u32, out: *mut *mut zkw_delivery) -> c_int;
    pub fn zkw_delivery_destroy(d: *mut zkw_delivery);
    pub fn zkw_delivery_slot_bytes(batches: *const *mut zkw_batch, n_batches: u32, worst_case: *mut u64) -> c_int;
    pub fn zkw_delivery_submit(d: *mut zkw_delivery, batches: *const *mut zkw_batch, n_batches: u32, run_stream: *mut c_void, ticket: *mut u32) -> c_int;
    pub fn zkw_delivery_order_after(d: *mut zkw_delivery, ticket: u32, stream: *mut c_void) -> c_int;
    pub fn zkw_delivery_wait(d: *mut zkw_delivery, ticket: u32, info: *mut zkw_delivered) -> c_int;
    pub fn zkw_delivery_get_instance_trace(d: *mut zkw_delivery, ticket: u32, batch_index: u32, instance: u32, out: *mut zkw_instance_trace) -> c_int;
    pub fn zkw_delivery_replay(d: *mut zkw_delivery, ticket: u32, f: Option<zkw_cycle_fn>, user: *mut c_void, n_cycles: *mut u64, checksum: *mut u64) -> c_int;
    pub fn zkw_delivery_release(d: *mut zkw_delivery, ticket: u32) -> c_int;
    // fresh inputs of an uploaded batch
    pub fn zkw_batch_staging(batch: *mut zkw_batch, states: *mut *mut zkw_vm_local_state, heap_words: *mut *mut zkw_u256, n_heap_words: *mut u32) -> c_int;
    pub fn zkw_batch_restage(batch: *mut zkw_batch, states: *const zkw_vm_local_state, heap_words: *const zkw_u256, n_heap_words: u32, stream: *mut c_void) -> c_int;
    pub fn zkw_blake2s256(ctx: *mut zkw_ctx, data: *const u8, offsets: *const u64, n_messages: u32, digests: *mut u8) -> c_int;
    pub fn zkw_blake2s256_device(ctx: *mut zkw_ctx, d_data: *const c_void, total_bytes: u64, d_offsets: *const u64, n_messages: u32,
                                 d_digests: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkw_abi_sizeof(which: u32) -> u32;
}

/// the layouts above against the library that was actually linked
pub fn check_abi() {
    use std::mem::size_of;
    assert_eq!(size_of::<zkw_callstack_entry>(), 112);
    assert_eq!(size_of::<zkw_vm_local_state>(), 680);
    assert_eq!(size_of::<zkw_cycle_record>(), 512);
    assert_eq!(size_of::<zkw_mem_query>(), 48);
    assert_eq!(size_of::<zkw_log_query>(), 128);
    assert_eq!(size_of::<zkw_aux_event>(), 256);
    assert_eq!(unsafe { zkw_abi_sizeof(6) }, 512);
}
