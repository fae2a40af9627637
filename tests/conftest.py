import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cp():
    import cpamd
    return cpamd.load()


@pytest.fixture(scope="session")
def orc():
    import orc_binding
    return orc_binding.OracleBackend()


@pytest.fixture(scope="session")
def hip(cp):
    """The product backend (HIP C-ABI library).  GPU tests only."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return cp.get_backend()
