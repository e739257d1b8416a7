"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle on the
same seeded inputs — bit-exact over every record of every instance."""
import numpy as np
import pytest

from era_zk_evm_amd import capi as K, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def product(isa):
    be = K.load_product().open(isa)
    yield be
    be.close()


def _run(backend, wl, lanes=0):
    wl.limits["lanes_per_wave"] = lanes
    b = backend.create_batch(wl)
    b.reset()
    b.run(wl.n_cycles)
    b.sync()
    return b


def _compare(oracle, product, wl, lanes=0, sample=None):
    bo = _run(oracle, wl)
    bp = _run(product, wl, lanes)
    idx = range(wl.n_instances) if sample is None else sample
    for i in idx:
        ok, why = K.traces_equal(bo.trace(i), bp.trace(i))
        assert ok, "%s instance %d (lanes=%d): %s" % (wl.name, i, lanes, why)
    assert int(bp.stats()["cycles"]) == int(bo.stats()["cycles"])
    bo.destroy()
    bp.destroy()


def test_cfg0_nop_add(oracle, product, isa):
    _compare(oracle, product, synth.make(0, isa))


@pytest.mark.parametrize("lanes", [0, 1, 16, 64])
def test_cfg1_arith_256x256(oracle, product, isa, lanes):
    _compare(oracle, product, synth.make(1, isa), lanes)


@pytest.mark.parametrize("lanes", [0, 8, 64])
def test_cfg2_mixed(oracle, product, isa, lanes):
    _compare(oracle, product, synth.make(2, isa, n_instances=320), lanes)


def test_cfg2_ragged_last_wave(oracle, product, isa):
    # instance count that is not a multiple of the lanes per wave
    _compare(oracle, product, synth.make(2, isa, n_instances=77), 64)


def test_cfg2_full_size_sampled(oracle, product, isa):
    # BASELINE config: 4096 x 256 = 1M cycles; every 37th instance compared record by record
    wl = synth.make(2, isa, n_instances=4096)
    _compare(oracle, product, wl, 0, sample=range(0, 4096, 37))


def test_rerun_after_reset_is_identical(product, isa):
    wl = synth.make(2, isa, n_instances=64)
    b = product.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    t1 = [b.trace(i) for i in (0, 17, 63)]
    b.reset(); b.run(wl.n_cycles); b.sync()
    t2 = [b.trace(i) for i in (0, 17, 63)]
    for a, c in zip(t1, t2):
        ok, why = K.traces_equal(a, c)
        assert ok, why


def test_split_run_equals_single_run(product, isa):
    wl = synth.make(2, isa, n_instances=64)
    b = product.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    whole = [b.trace(i) for i in (0, 31)]
    b.reset(); b.run(100); b.run(wl.n_cycles - 100); b.sync()
    parts = [b.trace(i) for i in (0, 31)]
    for a, c in zip(whole, parts):
        ok, why = K.traces_equal(a, c)
        assert ok, why
