import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first(request):
    """GPU runs only.  A few `-m gpu` tests hand torch streams / tensors to libzkw.so (the way bench.py does).  bench.py
    initialises torch's HIP runtime before it loads libzkw.so; the other order — libzkw.so first, torch.cuda later in the
    same process — was seen to end in "no ROCm-capable device is detected" inside torch depending on which tests ran
    before.  So the session brings torch's runtime up first, exactly like bench.py."""
    markexpr = request.config.getoption("markexpr", "") or ""
    if "not gpu" in markexpr:
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # no torch / no GPU: the tests that need them fail or skip on their own
        pass


@pytest.fixture(scope="session")
def isa():
    from era_zk_evm_amd import capi
    return capi.Isa()


@pytest.fixture(scope="session")
def oracle(isa):
    """TEST INFRASTRUCTURE: the CPU restatement of the reference (oracle/)."""
    from _oracle import load_oracle
    be = load_oracle().open(isa)
    yield be
    be.close()
