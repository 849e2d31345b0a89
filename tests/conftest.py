import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only).  Built on demand with g++."""
    from oracle import kvc_oracle
    kvc_oracle.build()
    kvc_oracle.lib()
    return kvc_oracle


@pytest.fixture(scope="session")
def kvc():
    """The product's ctypes binding of libkvc_hip.so."""
    from kvcache_factory_amd import _kvc
    _kvc.lib()
    return _kvc


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return torch.device("cuda:0")
