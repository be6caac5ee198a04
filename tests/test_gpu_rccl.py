"""The RCCL binding of libmgrt.so as far as one GPU can exercise it: a communicator of one rank."""
import pytest

torch = pytest.importorskip("torch")
from multigrid_amd import _lib

pytestmark = pytest.mark.gpu


def test_rccl_operations_direct_and_from_a_graph():
    # send/receive with itself in one group, all-gather, all-reduce through the dlopen'ed symbols (id by
    # value, enum values as in rccl.h), then the same captured into a hipGraph and replayed twice on
    # fresh data: what mgrt_slab_plan does per cycle with more ranks
    _lib.init(0)
    stream = torch.cuda.Stream()
    _lib.rccl_selftest(stream)
    torch.cuda.synchronize()
    # and the library still works afterwards
    assert len(_lib.rccl_unique_id()) == 128
