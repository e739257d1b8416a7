import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


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
