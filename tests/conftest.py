import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _oracle_up_to_date(target):
    """Where the reference's sources are at hand (the build container) the oracle libraries follow their
    sources by make, so that a changed .inc never meets a stale library; elsewhere the libraries travel
    prebuilt and only a missing restatement is built (plain gcc, seconds)."""
    odir = os.path.join(ROOT, "oracle")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", odir] + ([target] if target else []))
    elif not os.path.exists(os.path.join(odir, "libmg_oracle.so")):
        subprocess.check_call(["make", "-C", odir])


@pytest.fixture(scope="session")
def mgo():
    """The C restatement oracle."""
    import _oracle
    _oracle_up_to_date(None)
    return _oracle.load("mgo")


@pytest.fixture(scope="session")
def mgr():
    """The reference's own kernels compiled for x86-64, or None where it was not built."""
    import _oracle
    if os.path.isdir("/root/reference"):
        _oracle_up_to_date("ref")
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
