"""CPU tests of the oracle: the C restatement against the golden vectors generated from the
reference's own kernels, against those kernels directly where they were built, and against
mathematical properties of the operators."""
import glob
import os

import numpy as np
import pytest

import _oracle

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def bits_equal(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def run_case(lib, g):
    """Replay a golden case with `lib`, returning (history, u)."""
    f = g["func"]
    dt = f.dtype.type
    shape = tuple(int(s) for s in g["shape"])
    zfull, a1, a2, v, cycles = (int(g[k]) for k in ("zfull", "a1", "a2", "v", "cycles"))
    omega = dt(float(g["omega"]))
    cell = lib.box_cells(shape)
    res = np.zeros(shape, dt)
    hist = []
    if str(g["mode"]) == "mg":
        u = np.zeros(shape, dt)
        lib.residual(cell, res, u, f)
        hist.append(lib.l2norm_exact(res))
        for _ in range(cycles):
            lib.mg(u, f, omega, a1, a2, v, 1, zfull)
            lib.residual(cell, res, u, f)
            hist.append(lib.l2norm_exact(res))
    else:
        u = lib.fmg(f, omega, a1, a2, v, cycles, zfull)
        lib.residual(cell, res, u, f)
        hist.append(lib.l2norm_exact(res))
    return np.array(hist), u, res


def test_golden_present():
    assert len(GOLDEN) >= 30


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_restatement_matches_golden(mgo, path):
    g = np.load(path)
    hist, u, res = run_case(mgo, g)
    assert bits_equal(u, g["u"])
    assert np.array_equal(hist, g["history"])
    # the reference's lossy tree norms of the final residual (what `solve` prints)
    assert mgo.l2norm_tree(res) == g["l2res_tree"]
    assert mgo.linfnorm_tree(res) == g["linfres_tree"]


@pytest.mark.parametrize("path", [p for p in GOLDEN if "_v22" in p and ("_17_" in p or "_33_" in p)],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_problem_setup_matches_golden(mgo, path):
    g = np.load(path)
    shape = tuple(int(s) for s in g["shape"])
    f, _ = mgo.discretize(int(g["problem"]), shape, g["func"].dtype.type)
    assert bits_equal(f, g["func"])


def test_known_answers_2d_257(mgo):
    """Known answers: the residual history of the 2D 257^2 Dirichlet V(2,2) float case (the reference's kernels run serially
    on the CPU before this tree existed), first six entries."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "2d_257_f32_v22.npz"))
    expect = [8.282931639e+01, 2.989080772e+00, 7.839347357e-02, 2.326151235e-03, 9.590241536e-05, 4.229104532e-05]
    assert np.allclose(g["history"][:6], expect, rtol=5e-10, atol=0)
    hist, _, _ = run_case(mgo, g)
    assert np.allclose(hist[:6], expect, rtol=5e-10, atol=0)


@pytest.mark.parametrize("shape", [(9, 17), (5, 9, 17)], ids=["2D", "3D"])
def test_box_rule_with_face_types_equals_the_painted_map(mgo, shape):
    """set_box_faces (the rule the cycle drivers use for the benchmark's Neumann configurations) against
    box_cells, which paints the faces in the order of genBuffer (rectangularborderhandler.cpp:59-62,103-108):
    same bytes, shared edges included."""
    faces = (_oracle.DIRICHLET, _oracle.NEUMANN, _oracle.NEUMANN, _oracle.DIRICHLET, _oracle.DIRICHLET, _oracle.NEUMANN)
    mgo.set_box_faces(faces)
    try:
        got = mgo.level_cells(shape)
    finally:
        mgo.set_box_faces(None)
    assert np.array_equal(got, mgo.box_cells(shape, faces))
    assert np.array_equal(mgo.level_cells(shape), mgo.box_cells(shape))


def random_cells(lib, rng, shape):
    cell = lib.box_cells(shape)
    inner = cell[(slice(2, -2),) * len(shape)]
    m = rng.random(inner.shape)
    inner[m < 0.1] = _oracle.NEUMANN
    inner[(m >= 0.1) & (m < 0.15)] = _oracle.DIRICHLET
    inner[(m >= 0.15) & (m < 0.2)] = _oracle.OUTER
    return cell


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(9, 11, 13), (17, 17), (5, 5, 5), (7, 9), (9, 9, 17)])
def test_restatement_matches_reference_kernels(mgo, mgr, dt, shape):
    if mgr is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(hash((shape, dt.__name__)) % 2**32)
    dim = len(shape)
    for cell in (mgo.box_cells(shape), random_cells(mgo, rng, shape),
                 mgo.box_cells(shape, (2, 3, 3, 2, 2, 3))):
        u = rng.standard_normal(shape).astype(dt)
        f = rng.standard_normal(shape).astype(dt)
        zc = None if dim == 2 else (shape[0] + 1) // 2
        for odd in (0, 1):
            for zcount in (zc, None):
                a, b = u.copy(), u.copy()
                mgo.smooth_pass(cell, a, f, dt(0.8), odd, zcount)
                mgr.smooth_pass(cell, b, f, dt(0.8), odd, zcount)
                assert bits_equal(a, b)
        a, b = np.full(shape, 7, dt), np.full(shape, 7, dt)
        mgo.residual(cell, a, u, f)
        mgr.residual(cell, b, u, f)
        assert bits_equal(a, b)
        cs = tuple((s + 1) // 2 for s in shape)
        a, b = np.zeros(cs, dt), np.zeros(cs, dt)
        mgo.restrict(cell, a, u)
        mgr.restrict(cell, b, u)
        assert bits_equal(a, b)
        e = rng.standard_normal(cs).astype(dt)
        a, b = np.zeros(shape, dt), np.zeros(shape, dt)
        mgo.correct(cell, a, u, e)
        mgr.correct(cell, b, u, e)
        assert bits_equal(a, b)
        mgo.prolong(cell, a, e)
        mgr.prolong(cell, b, e)
        assert bits_equal(a, b)
        a, b = u.copy(), u.copy()
        mgo.zero_out(cell, a)
        mgr.zero_out(cell, b)
        assert bits_equal(a, b)
        assert mgo.l2norm_tree(u) == mgr.l2norm_tree(u)
        assert mgo.linfnorm_tree(u) == mgr.linfnorm_tree(u)
        assert bits_equal(mgo.diff(u, f), mgr.diff(u, f))
        assert bits_equal(mgo.scale(u, 0.3), mgr.scale(u, 0.3))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 17, 289, 4913])
def test_tree_norm_drops_tail(mgo, n):
    """auxiliary.cpp:119-135: elements past 4*(size/4) are lost at every level."""
    x = np.arange(1, n + 1, dtype=np.float64)
    size, vals = n, x * x
    while size != 1:
        new = max(1, size // 4)
        vals = np.array([vals[4 * i:min(size, 4 * i + 4)].sum() for i in range(new)])
        size = new
    assert mgo.l2norm_tree(x) == np.sqrt(vals[0])
    if n > 7:
        assert mgo.l2norm_tree(x) < mgo.l2norm_exact(x)


@pytest.mark.parametrize("shape", [(9, 9, 9), (17, 17)])
def test_operator_properties(mgo, shape):
    dt = np.float64
    rng = np.random.default_rng(7)
    cell = mgo.box_cells(shape)
    dim = len(shape)
    cs = tuple((s + 1) // 2 for s in shape)
    # full weighting keeps constants on interior coarse points (weights sum to 1)
    c = np.zeros(cs, dt)
    mgo.restrict(cell, c, np.full(shape, 3.0, dt))
    assert np.all(c == 3.0)
    # interpolation reproduces (tri)linear functions
    idx = np.meshgrid(*[np.arange(s, dtype=dt) for s in cs], indexing="ij")
    coef = rng.standard_normal(dim)
    coarse = sum(k * i for k, i in zip(coef, idx)) + 0.5
    fine_idx = np.meshgrid(*[np.arange(s, dtype=dt) / 2 for s in shape], indexing="ij")
    expect = sum(k * i for k, i in zip(coef, fine_idx)) + 0.5
    out = np.zeros(shape, dt)
    mgo.prolong(cell, out, coarse)
    inner = (slice(1, -1),) * dim  # border cells take the coarse value at (x/2,y/2,z/2) instead
    assert np.allclose(out[inner], expect[inner], rtol=0, atol=1e-13)
    # residual of f := A u vanishes on inner cells; smoothing a solution leaves it a solution
    u = rng.standard_normal(shape)
    f = np.zeros(shape, dt)
    lap = -2 * dim * u[inner]
    for ax in range(dim):
        lo = [slice(1, -1)] * dim
        hi = [slice(1, -1)] * dim
        lo[ax], hi[ax] = slice(0, -2), slice(2, None)
        lap = lap + u[tuple(lo)] + u[tuple(hi)]
    f[inner] = lap
    border = cell == _oracle.DIRICHLET
    f[border] = u[border]
    r = np.zeros(shape, dt)
    mgo.residual(cell, r, u, f)
    assert np.abs(r).max() < 1e-13
    u2 = u.copy()
    mgo.smoother_iterate(u2, f, 1.0, 2, zfull=True)
    assert np.abs(u2 - u).max() < 1e-13


def test_discretisation_error_is_second_order(mgo):
    """Solving to the floor on 9^3/17^3/33^3 (double, full-z smoothing): error vs tridimS1 ~ h^2."""
    errs = []
    for n in (9, 17, 33):
        shape = (n, n, n)
        f, sol = mgo.discretize(0, shape, np.float64)
        u = np.zeros(shape)
        mgo.mg(u, f, 1.0, 2, 2, 1, 12, True)
        errs.append(np.abs(u - sol).max())
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5
