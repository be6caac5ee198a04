"""mgrt_sumsq_noalloc_f32, the entry point that allocates nothing, against mgrt_sumsq_f32 through the
C ABI.  Both sum the same squares in double; only the grouping of the block partials can differ, so
the tolerance is 1e-13 relative (and equality where the grouping is the same)."""
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_sumsq_matches_sumsq_twice_in_a_row(dev):
    import torch
    from multigrid_amd import _lib as L
    torch.manual_seed(17)
    # second call on a smaller input: the shared array still holds the partials of the first
    for shape, rows in (((257, 300, 513), (3, 290)), ((33, 65, 129), (0, 65)), ((65, 129), (1, 64))):
        x = torch.randn(shape, device=dev)
        g = L.grid(shape)
        want = torch.zeros(1, dtype=torch.float64, device=dev)
        got = torch.full((1,), -1.0, dtype=torch.float64, device=dev)
        L.sumsq(g, x, want, *rows)
        L.sumsq_noalloc(g, x, got, *rows)
        torch.cuda.synchronize()
        w = want.item()
        ref = float((x.double()[..., rows[0]:rows[1], :] ** 2).sum())
        assert abs(got.item() - w) <= 1e-13 * w, (shape, got.item(), w)
        assert abs(w - ref) <= 1e-12 * ref
    # an empty row range gives 0, and a plane beyond the fixed array's reach is refused with a message
    L.sumsq_noalloc(g, x, got, 5, 5)
    torch.cuda.synchronize()
    assert got.item() == 0.0
    big = L.grid((4097, 4097))
    with pytest.raises(L.MgrtError, match="fixed scratch array"):
        L.sumsq_noalloc(big, x, got)
