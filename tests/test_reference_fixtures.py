"""Parity against the REFERENCE ITSELF: fixtures written by rust/zkw-refdump (the real zk_evm v1.4.1 `VmState::cycle`
with a recording `VmWitnessTracer`, tapes encoded with the real `zkevm_opcode_defs` table).  They can only be generated
on a machine with cargo + network (recipe: rust/zkw-refdump/Cargo.toml); when tests/golden/ref_*.bin are absent the
reference cases are skipped and only the loader's self-check runs (the same files fabricated from the oracle with the
build's default table, in a scratch directory).  With the fixtures present: the oracle (always) and the HIP path
(`-m gpu`) must reproduce every record, query, event and final state of the reference bit for bit."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from era_zk_evm_amd import capi as K

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, GOLD)
import ref_container as RC  # noqa: E402

FIXTURES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLD, "ref_*.bin"))
                  if not os.path.basename(p).startswith("ref_inputs_") and os.path.basename(p) != "ref_isa.bin")


def _norm_aux(aux):
    """what the reference cannot know is masked: the device's bookkeeping in a DECOMMIT event (code blob id in the upper
    half of `c`, preimage index behind the hash)"""
    a = aux.copy()
    dec = a["type"] == K.AUX_DECOMMIT
    a["c"][dec] &= 0xFFFF
    a["raw"][dec, 32:] = 0
    return a


def _equal(ref, got):
    for k in ("status", "n_cycles"):
        if ref[k] != got[k]:
            return False, "%s: %r != %r" % (k, ref[k], got[k])
    for k in ("records", "mem", "log", "mem_off", "log_off", "aux_off"):
        if np.ascontiguousarray(ref[k]).tobytes() != np.ascontiguousarray(got[k]).tobytes():
            return False, k
    if _norm_aux(ref["aux"]).tobytes() != _norm_aux(got["aux"]).tobytes():
        return False, "aux"
    if ref["final_state"].tobytes() != got["final_state"].tobytes():
        return False, "final_state"
    return True, ""


def _check(backend_factory, directory, name):
    isa = K.Isa(table=np.frombuffer(RC.read_container(os.path.join(directory, "ref_isa.bin"))["isa"], dtype=K.ISA_TABLE).copy())
    wl = RC.workload_from_sections(RC.read_container(os.path.join(directory, "ref_inputs_%s.bin" % name)))
    ref = RC.read_container(os.path.join(directory, "ref_%s.bin" % name))
    be = backend_factory().open(isa)
    b = be.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    for i in range(wl.n_instances):
        ok, why = _equal(RC.reference_trace(ref, i), b.trace(i))
        assert ok, "%s instance %d differs from the reference: %s" % (name, i, why)
    be.close()


@pytest.mark.skipif(not FIXTURES, reason="tests/golden/ref_*.bin absent: generate them with rust/zkw-refdump on a machine with cargo")
@pytest.mark.parametrize("name", FIXTURES or ["none"])
def test_oracle_reproduces_the_reference(name):
    from _oracle import load_oracle
    _check(load_oracle, GOLD, name)


@pytest.mark.gpu
@pytest.mark.skipif(not FIXTURES, reason="tests/golden/ref_*.bin absent: generate them with rust/zkw-refdump on a machine with cargo")
@pytest.mark.parametrize("name", FIXTURES or ["none"])
def test_hip_path_reproduces_the_reference(name):
    _check(K.load_product, GOLD, name)


def test_fixture_loader_self_check(tmp_path):
    """the plumbing (container format, workload round trip, trace comparison) on files fabricated from the oracle"""
    from _oracle import load_oracle
    r = subprocess.run([sys.executable, os.path.join(GOLD, "make_ref_inputs.py"), "--self-check", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    for name in ("cfg0", "cfg2", "cfg3", "cfg4", "fuzz"):
        _check(load_oracle, str(tmp_path), name)
