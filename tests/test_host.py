"""The C++ host classes (multigrid_amd/host): Buffer, BorderHandler, FunctionHandler, MultigridSolver,
the norms and the multigrid2D / multigrid3D executables, driven as a user of the reference would
drive them and compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "multigrid_amd", "host")
BINARIES = ["multigrid2D", "multigrid3D", "multigrid2D_d", "multigrid3D_d", "host_check", "host_check_d"]


def have(name):
    return os.access(os.path.join(HOST, name), os.X_OK)


@pytest.mark.parametrize("name", BINARIES)
def test_host_binaries_are_built(name):
    assert have(name), "run `make host` (multigrid_amd/host/%s is missing)" % name


def test_no_device_is_reported_not_worked_around():
    """Without a GPU the executables must stop with the reference's message
    (clcontextloader.cpp:56), not compute anything on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([os.path.join(HOST, "multigrid3D"), "--profiling"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1
    assert "No devices found" in p.stdout


@pytest.mark.parametrize("extents, want", [
    ((257, 513, 257), (257, 3, 1)),     # the 513^3 smoother range (multigridsolver3D.cpp:121): 257 is prime
    ((513, 513, 513), (513, 1, 1)),
    ((129, 257), (129, 1)),
    ((64, 64, 64), (64, 16, 1)),
    ((4097,), (241,)),                  # 4097 = 17 * 241
    ((2053, 8), (1, 8)),                # a prime beyond the per-dimension limit
])
def test_get_best_workspace_dim(extents, want):
    """getBestWorkspaceDim (auxiliary.cpp:33-117): largest divisors within 1024 per dimension, first
    dimension first, product within 1024."""
    exe = os.path.join(HOST, "workspace_dim")
    assert os.access(exe, os.X_OK), "run `make host`"
    out = subprocess.run([exe] + [str(e) for e in extents], capture_output=True, text=True, timeout=30)
    got = tuple(int(t) for t in out.stdout.split())
    assert got == want
    prod = 1
    for e, g in zip(extents, got):
        assert e % g == 0
        prod *= g
    assert prod <= 1024


def celltype(x, y, z, nx, ny, nz):
    threed = nz > 1
    if x == 0 or x == nx - 1:
        return _oracle.DIRICHLET
    if y == 0 or y == ny - 1 or (threed and z == 0):
        return _oracle.NEUMANN
    if threed and z == nz - 1:
        return _oracle.DIRICHLET
    r2 = (x - nx // 2) ** 2 + (y - ny // 2) ** 2
    if nx > 5 and r2 < (nx // 5) ** 2:
        return _oracle.OUTER if r2 < (nx // 8) ** 2 else _oracle.DIRICHLET
    return _oracle.INNER


def run_check(tmp_path, dt, dim, n, mode, omega, a1, a2, v, iters, fullz, domain):
    exe = os.path.join(HOST, "host_check" if dt == np.float32 else "host_check_d")
    out = str(tmp_path / "out.bin")
    nx, ny, nz = n
    cmd = [exe, out, str(dim), str(nx), str(ny), str(nz), mode, repr(float(omega)), str(a1), str(a2), str(v),
           str(iters), str(int(fullz)), domain]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    raw = open(out, "rb").read()
    hx, hy, hz, w = struct.unpack("4i", raw[:16])
    assert (hx, hy, hz, w) == (nx, ny, nz if dim == 3 else 1, np.dtype(dt).itemsize)
    cnt = hx * hy * hz
    shape = (hz, hy, hx) if dim == 3 else (hy, hx)
    arrs = np.frombuffer(raw, dtype=dt, count=4 * cnt, offset=16).reshape((4,) + shape)
    norms = np.frombuffer(raw, dtype=np.float64, count=5, offset=16 + 4 * cnt * w)
    return arrs[0], arrs[1], arrs[2], arrs[3], norms, p.stdout


def oracle_solve(mgo, dt, dim, n, mode, omega, a1, a2, v, iters, fullz):
    nx, ny, nz = n
    shape = (nz, ny, nx) if dim == 3 else (ny, nx)
    f, sol = mgo.discretize(0 if dim == 3 else 2, shape, dt)
    u = np.zeros(shape, dt)
    if mode.startswith("slab"):
        mode = mode[5:]  # the same solve, whoever computes it
    if mode in ("mg", "debug"):
        mgo.mg(u, f, dt(omega), a1, a2, v, iters, fullz)
    elif mode == "fmg":
        u = mgo.fmg(f, dt(omega), a1, a2, v, iters, fullz)
    elif mode == "sor":
        mgo.smoother_iterate(u, f, dt(omega), a1, fullz)
    elif mode == "steps":
        cshape = tuple((s + 1) // 2 for s in shape)
        cell, ccell = mgo.level_cells(shape), mgo.level_cells(cshape)
        tmp, rc, e = np.zeros(shape, dt), np.zeros(cshape, dt), np.zeros(cshape, dt)
        mgo.smoother_iterate(u, f, dt(omega), a1, fullz)
        mgo.residual(cell, tmp, u, f)
        mgo.restrict(cell, rc, tmp)
        mgo.smoother_iterate(e, rc, dt(omega), a1 + a2, fullz)
        out = np.zeros(shape, dt)
        mgo.correct(cell, out, u, e)
        u = out
        mgo.smoother_iterate(u, f, dt(omega), a2, fullz)
    cell = mgo.level_cells(shape)
    mgo.zero_out(cell, u)
    err = mgo.diff(u, sol)
    r = np.zeros(shape, dt)
    mgo.residual(cell, r, u, f)
    return f, u, r, err


CASES = [
    # dim, (nx, ny, nz), mode, omega, a1, a2, v, iters, fullz, domain
    (3, (17, 17, 17), "mg", 1.0, 2, 2, 1, 3, 0, "box"),
    (3, (17, 17, 17), "mg", 1.0, 2, 2, 1, 3, 1, "box"),
    (3, (33, 17, 9), "fmg", 0.9, 3, 3, 2, 3, 1, "box"),
    (3, (17, 33, 17), "mg", 0.9, 2, 1, 2, 2, 1, "map"),
    (3, (17, 33, 17), "fmg", 1.0, 1, 2, 1, 2, 0, "map"),
    (3, (18, 17, 21), "mg", 1.0, 2, 2, 1, 2, 1, "box"),      # an even extent: the step-by-step path, runs only
    (3, (33, 33, 33), "steps", 0.8, 2, 2, 1, 1, 0, "box"),
    (3, (9, 9, 9), "sor", 1.2, 5, 0, 1, 1, 0, "box"),
    # MultigridSolver3D::distribute: 2 or 3 solvers, each a thread holding its rows only
    (3, (33, 65, 17), "slab2mg", 1.0, 2, 2, 1, 2, 1, "box"),
    (3, (17, 129, 33), "slab3mg", 0.9, 3, 1, 2, 1, 0, "box"),
    (3, (33, 65, 33), "slab2fmg", 1.0, 2, 2, 1, 2, 1, "box"),
    (2, (33, 33, 1), "mg", 1.0, 2, 2, 1, 4, 0, "box"),
    (2, (65, 33, 1), "fmg", 2.0 / 3.0, 3, 3, 2, 3, 0, "box"),
    (2, (33, 65, 1), "mg", 0.9, 2, 2, 2, 2, 0, "map"),
    (2, (34, 33, 1), "mg", 1.0, 2, 2, 1, 2, 0, "box"),
    (2, (33, 33, 1), "debug", 1.0, 2, 2, 1, 1, 0, "box"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dD-%s-%s-%s" % (c[0], "x".join(map(str, c[1])), c[2], c[9]))
def test_host_classes_match_oracle(mgo, tmp_path, dt, case):
    dim, n, mode, omega, a1, a2, v, iters, fullz, domain = case
    f, u, r, err, norms, stdout = run_check(tmp_path, dt, dim, n, mode, omega, a1, a2, v, iters, fullz, domain)
    mgo.set_celltype(celltype if domain == "map" else None)
    try:
        rf, ru, rr, rerr = oracle_solve(mgo, dt, dim, n, mode, dt(omega), a1, a2, v, iters, bool(fullz))
        tree = (mgo.l2norm_tree(rr), mgo.linfnorm_tree(rr))
    finally:
        mgo.set_celltype(None)
    assert np.array_equal(f.view(np.uint8), rf.view(np.uint8)), "discretize_func differs"
    if any(e % 2 == 0 for e in n[:dim]):
        # Even extents: the reference's interpolation reads one element past the coarse array
        # (mg_1.cl:222-229 with x = nx-1 odd), so its results are undefined and there is nothing to
        # compare with.  The classes must still run the cycle step by step, with those indices
        # clamped, rather than stop.
        return
    tol = 1e-5 if dt == np.float32 else 1e-12
    scale = max(1.0, float(np.abs(ru).max()))
    assert np.abs(u.astype(np.float64) - ru).max() <= tol * scale
    if mode.startswith("slab"):
        assert np.array_equal(u.view(np.uint8), ru.view(np.uint8)), "the distributed solve differs from the whole-grid one"
    assert np.abs(r.astype(np.float64) - rr).max() <= tol * max(1.0, float(np.abs(rr).max())) * 8
    assert np.abs(err.astype(np.float64) - rerr).max() <= tol * scale
    if np.array_equal(r.view(np.uint8), rr.view(np.uint8)):
        # same residual bits -> the reference-tree norms must agree to the last bit
        assert norms[0] == tree[0] and norms[1] == tree[1]
    ex = float(np.sqrt((rr.astype(np.float64) ** 2).sum()))
    assert abs(norms[3] - ex) <= 1e-4 * max(ex, 1e-30)
    if mode == "debug":
        assert stdout.count("DEBUG, printing residuals on grid: 33x33 k= 0 BEFORE correction res= ") == 1
        assert stdout.count("AFTER correction") == 4  # levels 33, 17, 9, 5


def repl(exe, script, *args):
    p = subprocess.run([os.path.join(HOST, exe)] + list(args), input=script, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


@pytest.mark.gpu
def test_repl_session_3d(mgo, tmp_path):
    """A scripted session of the 3D executable: the commands of programstate.cpp:36-53, the result
    table of :642-647, compared with the oracle's numbers at the printed precision."""
    sol = str(tmp_path / "sol.txt")
    script = "\n".join(["state", "setmode mg", "setdim 17 17 17", "setsmoothsteps 2 2", "setvcycles 1",
                        "setiterations 3", "setomega 1", "bogus", "setdim -1 2 3", "solve", "save sol " + sol,
                        "save res x.bmp", "reduce sol", "print nothing", "quit"]) + "\n"
    out = repl("multigrid3D", script)
    assert "Picking device: " in out
    assert "Real epsilon is: 1.19209e-07" in out
    assert "Current solver: FMG\nDimension: 17x17\nPre smooth steps: 3 Post smooth steps: 3\nVCycles: 2 Omega: 1\n" in out
    assert "Unknown command: bogus" in out and "Invalid dimensions" in out
    assert "Saving to bitmap not supported in 3D." in out and "Print what?" in out and out.rstrip().endswith("Quitting...")
    lines = out.split("\n")
    i = next(k for k, l in enumerate(lines) if l.endswith("Time\t\t\tL2Err\t\t\tLInfErr\t\t\tL2Res\t\t\tLinfRes\t\t\t"))
    cols = lines[i + 1].split("\t\t\t")
    shape = (17, 17, 17)
    f, exact = mgo.discretize(0, shape, np.float32)
    u = np.zeros(shape, np.float32)
    mgo.mg(u, f, np.float32(1), 2, 2, 1, 3, False)   # the reference's z range
    cell = mgo.level_cells(shape)
    r = np.zeros(shape, np.float32)
    mgo.residual(cell, r, u, f)
    e = mgo.diff(u, exact)
    want = [mgo.l2norm_tree(e), mgo.linfnorm_tree(e), mgo.l2norm_tree(r), mgo.linfnorm_tree(r)]
    for got, w in zip(cols[1:5], want):
        assert abs(float(got) - w) <= 2e-5 * abs(w) + 1e-12, (cols, want)
    # `save`: one text line per x, y along the line, a blank line between z planes (:135-148)
    rows = open(sol).read().split("\n")
    assert len(rows[0].split()) == 17 and rows[17] == ""
    saved = np.array([[float(t) for t in row.split()] for row in rows if row.strip()]).reshape(17, 17, 17)  # z, x, y
    assert np.abs(saved.transpose(0, 2, 1) - u).max() <= 1e-5 * max(1.0, np.abs(u).max())


@pytest.mark.gpu
def test_repl_profiling_2d_fmg_defaults(mgo):
    """`multigrid2D --profiling`: default FMG solve (a=3, v=2, iterations=3, 17x17, sinfunc2) and exit."""
    out = repl("multigrid2D", "", "--profiling")
    cols = out.strip().split("\n")[-1].split("\t\t\t")
    shape = (17, 17)
    f, exact = mgo.discretize(1, shape, np.float32)
    u = mgo.fmg(f, np.float32(1), 3, 3, 2, 3, False)
    r = np.zeros(shape, np.float32)
    mgo.residual(mgo.level_cells(shape), r, u, f)
    e = mgo.diff(u, exact)
    want = [mgo.l2norm_tree(e), mgo.linfnorm_tree(e), mgo.l2norm_tree(r), mgo.linfnorm_tree(r)]
    for got, w in zip(cols[1:5], want):
        assert abs(float(got) - w) <= 1e-4 * abs(w) + 1e-12, (cols, want)


@pytest.mark.gpu
def test_bitmap_roundtrip_2d(tmp_path):
    """save sol x.bmp writes an 8-bit palettised file (auxiliary.cpp:249-303): headers, grey
    palette, rows padded to 4."""
    bmp = str(tmp_path / "sol.bmp")
    repl("multigrid2D", "setdim 33 33\nsolve\nsave sol %s\nquit\n" % bmp)
    raw = open(bmp, "rb").read()
    assert raw[:2] == b"BM"
    size, _, _, off = struct.unpack("<IHHI", raw[2:14])
    hs, w, h, planes, bits = struct.unpack("<IiiHH", raw[14:30])
    assert (hs, w, h, planes, bits) == (40, 33, 33, 1, 8)
    assert off == 14 + 40 + 1024 and size == off + 36 * 33 == len(raw)
    assert raw[54:62] == bytes([0, 0, 0, 0, 1, 1, 1, 1])
    px = np.frombuffer(raw, np.uint8, offset=off).reshape(33, 36)
    assert px[:, 33:].max() == 0 and px.max() == 255


def test_bench_problem_of_a_slab_equals_the_rows_of_the_whole():
    # bench.py lets every rank sample its own rows; they must be the rows of the one-GPU problem
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for faces in ((2,) * 6, (2, 2, 3, 3, 2, 2)):
        for dt in (torch.float32, torch.float64):
            whole = bench.make_problem((9, 17, 9), dt, faces, torch.device("cpu"))
            for lo, hi in ((0, 7), (4, 13), (9, 17), (0, 17)):
                part = bench.make_problem((9, 17, 9), dt, faces, torch.device("cpu"), rows=(lo, hi))
                assert torch.equal(part, whole[:, lo:hi, :])


@pytest.mark.parametrize("shape, problem", [((33, 33, 33), 0), ((17, 33, 65), 0), ((65, 65), 1), ((33, 129), 1)])
@pytest.mark.parametrize("dtname", ["float32", "float64"])
def test_bench_problem_equals_the_oracles_discretisation(shape, problem, dtname):
    """bench.make_problem samples tridimF1 / sinfunc2 (programstate.cpp:84-94,114-122) as discretize_func
    does (functionhandler.cpp:25-45,64-84): bit for bit the oracle's right-hand side in both precisions."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench
    mgo = _oracle.load("mgo")
    npdt = np.dtype(dtname).type
    want, _ = mgo.discretize(problem, shape, npdt)
    got = bench.make_problem(shape, getattr(torch, dtname), (2,) * 6, torch.device("cpu")).numpy()
    bits = np.uint32 if npdt is np.float32 else np.uint64
    assert np.array_equal(got.view(bits), want.view(bits))
