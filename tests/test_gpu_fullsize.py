"""The configurations the benchmark is quoted on, at full size, through properties that need no
CPU reference: the default plan (fused sweeps, residual pass in the sweep, one-launch bottom) must
equal the unfused plan bit for bit -- the unfused kernels are the ones checked against the oracle
at small sizes --, the residual must fall at the known multigrid rate, and cells the reference never
writes must keep their values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def problem(shape, dtype, dev):
    """sin products sampled like discretize_func (functionhandler.cpp:64-84): f*h*h inside,
    the exact solution on the faces."""
    import math
    import torch
    axes = [torch.arange(n, device=dev, dtype=torch.float64) / (n - 1) for n in shape]
    s = [torch.sin(math.pi * a) for a in axes]
    if len(shape) == 3:
        sol = s[0][:, None, None] * s[1][None, :, None] * s[2][None, None, :]
        lap = -3.0 * math.pi ** 2
    else:
        sol = s[0][:, None] * s[1][None, :]
        lap = -2.0 * math.pi ** 2
    h = 1.0 / (shape[-1] - 1)
    f = (lap * sol * h * h).to(dtype)
    sol = sol.to(dtype)
    for d in range(len(shape)):
        idx = [slice(None)] * len(shape)
        for e in (0, -1):
            idx[d] = e
            f[tuple(idx)] = sol[tuple(idx)]
    return f.contiguous()


def run(shape, dtype, dev, cycles, faces=None, **kw):
    import torch
    from multigrid_amd import _lib as L
    f = problem(shape, dtype, dev)
    u = torch.zeros_like(f)
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())  # f and u are produced on the current stream
    with torch.cuda.stream(stream):
        plan = L.Plan(shape, dtype, zfull=True, **({"faces": faces} if faces else {}), **kw)
        hist = plan.vcycle(u, f, 1.0, 2, 2, 1, cycles, history=True)
        plan.close()
    torch.cuda.synchronize()
    return u, f, hist


def test_513_cubed_f32_fused_equals_unfused(dev):
    """The headline case on one GPU: 3D 513^3 Dirichlet, V(2,2)."""
    import torch
    shape = (513, 513, 513)
    u1, f, h1 = run(shape, torch.float32, dev, 2, use_graph=True)
    u2, _, h2 = run(shape, torch.float32, dev, 2, no_fuse=True)
    assert torch.equal(u1.view(torch.int32), u2.view(torch.int32))
    assert h1[0] == h2[0] and all(abs(a - b) <= 1e-9 * b for a, b in zip(h1, h2))
    assert h1[1] < 0.25 * h1[0] and h1[2] < 0.25 * h1[1]          # V(2,2) contracts by ~0.12 per cycle here
    assert torch.equal(u1[0], f[0]) and torch.equal(u1[:, :, -1], f[:, :, -1])  # Dirichlet faces hold f


def test_257_cubed_neumann_faces_are_frozen(dev):
    """3D 257^3 with Neumann faces, which the reference never updates, and the
    Dirichlet variant of the same grid."""
    import torch
    from multigrid_amd import _lib as L
    shape = (257, 257, 257)
    u1, f, h1 = run(shape, torch.float32, dev, 2, faces=(L.NEUMANN,) * 6, use_graph=True)
    u2, _, h2 = run(shape, torch.float32, dev, 2, faces=(L.NEUMANN,) * 6, no_fuse=True)
    assert torch.equal(u1.view(torch.int32), u2.view(torch.int32))
    for d in range(3):
        for e in (0, -1):
            idx = [slice(None)] * 3
            idx[d] = e
            assert not u1[tuple(idx)].any()          # started at 0 and must still be 0
    ud, fd, hd = run(shape, torch.float32, dev, 3, use_graph=True)
    assert hd[3] < 0.02 * hd[0]


def test_1025_cubed_f64_one_cycle(dev):
    """The largest case, on one GPU: 3D 1025^3 in double."""
    import torch
    shape = (1025, 1025, 1025)
    u1, f, h1 = run(shape, torch.float64, dev, 1, use_graph=True)
    ref = float(u1.double().abs().max())
    assert 0.3 < ref < 1.2 and h1[1] < 0.25 * h1[0]
    del f
    u2, _, h2 = run(shape, torch.float64, dev, 1, no_fuse=True)
    assert torch.equal(u1.view(torch.int64), u2.view(torch.int64))


def test_4097_squared_f32_mixed_faces(dev):
    """2D 4097^2, Dirichlet on the x faces, Neumann (frozen) on the y faces."""
    import torch
    from multigrid_amd import _lib as L
    shape = (4097, 4097)
    faces = (L.DIRICHLET, L.DIRICHLET, L.NEUMANN, L.NEUMANN, L.DIRICHLET, L.DIRICHLET)
    u, f, h = run(shape, torch.float32, dev, 3, faces=faces, use_graph=True)
    assert torch.equal(u[:, 0], f[:, 0]) and torch.equal(u[:, -1], f[:, -1])   # x faces: Dirichlet, u = f
    assert not u[0, 1:-1].any() and not u[-1, 1:-1].any()                      # y faces: never written
    assert h[-1] < h[0] and np.isfinite(h).all()


@pytest.mark.parametrize("case", [
    ((65, 513, 257), "f32", 0.9, 3, 1, 2, (2, 2, 2, 2, 2, 2)),
    ((129, 129, 1025), "f32", 1.0, 1, 2, 1, (2, 3, 3, 2, 2, 3)),
    ((257, 65, 513), "f64", 0.8, 2, 2, 1, (3, 2, 2, 1, 2, 2)),
    ((33, 1025, 129), "f64", 1.0, 2, 3, 2, (2, 2, 2, 2, 2, 2)),
], ids=lambda c: "x".join(map(str, c[0][::-1])) + "-" + c[1])
def test_plan_paths_agree_on_anisotropic_grids(dev, case):
    """Grids with unequal extents (levels stop when the shortest extent reaches 3), odd smoothing
    counts, W-cycles, mixed face classes: fused and unfused plans, bit for bit, on random data."""
    import torch
    from multigrid_amd import _lib as L
    shape, dt, omega, a1, a2, v, faces = case
    tdt = torch.float32 if dt == "f32" else torch.float64
    torch.manual_seed(7)
    f = torch.randn(shape, device=dev, dtype=tdt)
    u0 = torch.randn(shape, device=dev, dtype=tdt)
    outs = []
    for kw in (dict(use_graph=True), dict(no_fuse=True)):
        u = u0.clone()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            plan = L.Plan(shape, tdt, faces=faces, zfull=True, **kw)
            plan.vcycle(u, f, omega, a1, a2, v, 2)
            plan.close()
        torch.cuda.synchronize()
        outs.append(u)
    assert torch.equal(outs[0].view(torch.uint8), outs[1].view(torch.uint8))
    assert torch.isfinite(outs[0]).all()


def oracle_cycle(mgo, shape, dt, cycles=1):
    """V(2,2) cycles of the reference's 3D problem (tridimF1, programstate.cpp:114-122) from a zero
    guess on the CPU oracle, all z planes smoothed.  Returns (f, u)."""
    f, _ = mgo.discretize(0, shape, dt)
    u = np.zeros(shape, dt)
    mgo.mg(u, f, dt(1.0), 2, 2, 1, cycles, True)
    return f, u


def gpu_cycle(dev, f, cycles=1, maps=None, **kw):
    import torch
    from multigrid_amd import _lib as L
    df = torch.from_numpy(f).to(dev)
    du = torch.zeros_like(df)
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        plan = L.Plan(tuple(f.shape), df.dtype, zfull=True, **kw)
        if maps is not None:
            for l in range(plan.levels):
                plan.set_cells(l, maps(plan.level_shape(l)))
        plan.vcycle(du, df, 1.0, 2, 2, 1, cycles)
        plan.close()
    torch.cuda.synchronize()
    return du.cpu().numpy()


@pytest.mark.parametrize("dt", [np.float32, np.float64], ids=["f32", "f64"])
def test_257_cubed_cycle_matches_oracle(dev, mgo, dt):
    """One whole V(2,2) cycle at 257^3 (fused sweeps, graph replay, one-launch bottom) against the
    CPU oracle, bit for bit."""
    f, ref = oracle_cycle(mgo, (257, 257, 257), dt, cycles=2)
    got = gpu_cycle(dev, f, cycles=2, use_graph=True)
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))


def test_513_cubed_f32_cycle_matches_oracle(dev, mgo):
    """The benchmark's own workload: one V(2,2) cycle at 513^3 in float against the CPU oracle, bit
    for bit (this is the path with the residual pass and the correction inside the sweeps)."""
    f, ref = oracle_cycle(mgo, (513, 513, 513), np.float32)
    got = gpu_cycle(dev, f, use_graph=True)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_1025_cubed_f64_cycle_matches_oracle(dev, mgo):
    """The largest configuration at full size: one V(2,2) cycle at 1025^3 in double (planes with more tiles than
    the device runs at once, swept band by band; residual + restriction as one march) against the CPU oracle,
    bit for bit.  About 60 GB of host memory and half a minute."""
    f, ref = oracle_cycle(mgo, (1025, 1025, 1025), np.float64)
    got = gpu_cycle(dev, f, use_graph=True)
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))


@pytest.mark.parametrize("name", ["2d4097mixed", "3d257neumann", "2d257"])
def test_benchmark_configurations_with_their_faces_match_oracle(dev, mgo, name):
    """The benchmark's remaining workloads as bench.py sets them up (problem, faces, full z range), two V(2,2)
    cycles at full size against the CPU oracle, bit for bit: 4097^2 with Dirichlet x faces and frozen Neumann y
    faces, 257^3 with every face Neumann, 257^2."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    shape, dt, faces, problem, _ = bench.CONFIGS[name]
    npdt = np.float32 if dt == "f32" else np.float64
    mgo.set_box_faces(faces)
    try:
        f, _ = mgo.discretize(problem, shape, npdt)
        ref = np.zeros(shape, npdt)
        mgo.mg(ref, f, npdt(1.0), 2, 2, 1, 2, True)
    finally:
        mgo.set_box_faces(None)
    got = gpu_cycle(dev, f, cycles=2, use_graph=True, faces=faces)
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
    assert np.abs(ref).max() > 0


def test_257_cubed_ball_domain_matches_oracle(dev, mgo):
    """A domain that is not a box, at 257^3: the ball inscribed in the cube with a Dirichlet /
    Neumann shell and outer cells around it, described by one byte per point on every level."""
    shape = (257, 257, 257)
    mgo.set_builtin_celltype(1)
    try:
        f, ref = oracle_cycle(mgo, shape, np.float32)
        got = gpu_cycle(dev, f, maps=mgo.level_cells, use_graph=True)
    finally:
        mgo.set_builtin_celltype(0)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.abs(ref).max() > 0
