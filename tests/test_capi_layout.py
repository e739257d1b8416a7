"""The C-ABI library loads without a GPU and exports exactly what include/zkw.h declares; the numpy/ctypes
bindings of era-zk_evm_amd/capi.py have the struct sizes compiled into the library (zkw_abi_sizeof)."""
import ctypes as C
import os
import re

import pytest

from era_zk_evm_amd import build, capi as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "zkw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkw_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    build.build_lib()  # (re)built whenever a source is newer: hipcc's HOST pass of the kernels is part of what this suite checks
    return C.CDLL(build.LIB)  # dlopen only: no device call is made


def test_header_declares_the_documented_entry_points():
    names = _declared_functions()
    for must in ("zkw_ctx_create", "zkw_ctx_set_isa", "zkw_batch_create", "zkw_batch_set_state", "zkw_batch_upload", "zkw_batch_reset", "zkw_batch_run",
                 "zkw_batch_step", "zkw_batches_step", "zkw_batch_sync", "zkw_batch_get_instance_trace", "zkw_batch_commit", "zkw_batch_get_commitments", "zkw_batch_net_states", "zkw_batch_get_net_state",
                 "zkw_isa_default", "zkw_abi_sizeof"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_every_symbol_cited_in_integration_md_exists(lib):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    cited = sorted(set(re.findall(r"\b(zkw_(?:(?:ctx|batch|batches|isa|abi|comm|reduce|delivery)_[a-z0-9_]+|blake2s256(?:_device)?))\b", text)))
    types = {"zkw_isa_table", "zkw_isa_consts", "zkw_isa_entry", "zkw_comm_id"}  # struct names of include/zkw.h, not entry points
    assert cited
    missing = [n for n in cited if n not in types and not hasattr(lib, n)]
    assert not missing, missing


def test_struct_sizes_match_the_bindings(lib):
    lib.zkw_abi_sizeof.restype = C.c_uint32
    lib.zkw_abi_sizeof.argtypes = [C.c_uint32]
    expect = {0: K.ISA_TABLE.itemsize, 1: K.CALLSTACK_ENTRY.itemsize, 2: K.VM_LOCAL_STATE.itemsize, 3: K.BLOCK_PROPERTIES.itemsize,
              4: K.STORAGE_SLOT.itemsize, 5: K.LIMITS.itemsize, 6: K.CYCLE_RECORD.itemsize, 7: K.MEM_QUERY.itemsize, 8: K.LOG_QUERY.itemsize,
              9: K.AUX_EVENT.itemsize, 10: C.sizeof(K.InstanceTraceC), 11: K.RUN_STATS.itemsize, 12: K.ISA_CONSTS.itemsize,
              13: K.EVENT_MESSAGE.itemsize, 14: C.sizeof(K.NetStateC), 15: C.sizeof(K.DeliveredC)}
    for which, size in expect.items():
        assert lib.zkw_abi_sizeof(which) == size, which
    assert lib.zkw_abi_sizeof(99) == 0


def test_no_gpu_means_a_loud_error_not_a_fallback(lib):
    """Without a device zkw_ctx_create must fail (ZKW_ERR_DEVICE); with one it must succeed — never a CPU path."""
    import torch
    ctx = C.c_void_p()
    rc = lib.zkw_ctx_create(C.c_int(0), C.byref(ctx))
    if torch.cuda.is_available():
        assert rc == K.OK
        lib.zkw_ctx_destroy(ctx)
    else:
        assert rc == K.ERR_DEVICE
        assert not ctx.value
