"""What one GPU can tell about the several-GPU path: the RCCL binding on a communicator of one rank, and the
captured slab cycle against direct launches."""
import pytest

torch = pytest.importorskip("torch")
from multigrid_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backend", ["rccl", "rccl-nograph"])
@pytest.mark.parametrize("world", [2, 4])
def test_slab_plan_over_rccl_between_gpus_matches_the_oracle(world, backend):
    """mgrt_slab_plan with MGRT_TRANSPORT_RCCL on `world` GPUs, one process each: ncclSend/ncclRecv between
    neighbours, all-gather of the coarse rows, all-reduce of the norms, the cycle captured into a graph per
    rank (or launched directly).  Every rank's rows bit for bit the oracle's whole-grid cycle.  Needs that
    many GPUs: skipped on a one-GPU box, and the first thing to look at on a node."""
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, torch.cuda.device_count()))
    from test_dist import launch
    launch(world, backend, (33, 129, 65), "f32", 1.0, 2, 2, 1, 2, True, 9 ** 3, timeout=600)
    launch(world, backend, (17, 129, 33), "f64", 0.9, 2, 2, 1, 2, True, 9 ** 3, timeout=600)


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


@pytest.mark.parametrize("shape, dt", [((33, 129, 65), "f32"), ((17, 129, 33), "f64"), ((9, 1283, 513), "f32")],
                         ids=["small", "f64", "wide"])
def test_slab_cycle_replayed_from_a_graph_equals_direct_launches(shape, dt):
    """The share of a middle rank with the stand-alone transport (what it would receive is what it packed,
    copied on the device): the captured cycle -- kernels, halo kernels, copies, the side stream forked and
    joined by events -- must leave the same bits as the same cycle launched directly."""
    _lib.init(0)
    tdt = torch.float32 if dt == "f32" else torch.float64
    dev = torch.device("cuda", 0)
    outs = []
    for graph in (True, False):
        plan = _lib.SlabPlan(shape, tdt, transport=None, rank=1, nranks=4, zfull=True, crossover=9 ** 3, use_graph=graph)
        assert plan.distributed_levels >= 1
        lo, hi, _, _ = plan.rows()
        g = torch.Generator(device="cpu").manual_seed(5)
        f = torch.randn((shape[0], hi - lo, shape[2]), generator=g, dtype=tdt).to(dev)
        u = torch.randn((shape[0], hi - lo, shape[2]), generator=g, dtype=tdt).to(dev)
        hist = plan.vcycle(u, f, 1.0, 2, 2, 1, 3, history=True)
        torch.cuda.synchronize()
        assert bool(plan.uses_graph) == graph
        outs.append((u.cpu(), hist))
        plan.close()
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
