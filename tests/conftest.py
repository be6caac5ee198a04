import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mgo():
    """The C restatement oracle; built on demand (plain gcc, seconds)."""
    import _oracle
    if not os.path.exists(os.path.join(ROOT, "oracle", "libmg_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return _oracle.load("mgo")


@pytest.fixture(scope="session")
def mgr():
    """The reference's own kernels compiled for x86-64, or None where it was not built."""
    import _oracle
    return _oracle.load("mgr")


@pytest.fixture(scope="session")
def dev():
    """cuda:0 with libmgrt initialised on it."""
    import torch
    from multigrid_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    _lib.init(0)
    return torch.device("cuda:0")
