"""The slab-decomposed V-cycle (multigrid_amd/dist.py) with more than one process.

CPU: gloo, oracle-backed kernels, world sizes 2 and 3: partitioning, halo exchange schedule, coarse
gather.  GPU: two processes sharing cuda:0 (gloo with host staging), the HIP kernels on slabs.
Each rank checks its own rows bit for bit against the oracle's whole-grid cycle."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, backend, shape, dtype, omega, a1, a2, v, ncycles, zfull, crossover, timeout=300, faces=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "dist_worker.py"), backend] + [str(s) for s in shape] + \
          [dtype, repr(float(omega)), str(a1), str(a2), str(v), str(ncycles), str(int(zfull)), str(crossover)] + \
          ([",".join(map(str, faces))] if faces else [])
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if p.returncode != 0:
        sys.stderr.write("---- stdout ----\n%s\n---- stderr ----\n%s\n" % (p.stdout[-6000:], p.stderr[-6000:]))
    assert p.returncode == 0, "a rank failed: output is on stderr above"


def test_partition():
    from multigrid_amd import dist as slab
    nd, align = slab.plan_levels((513, 513, 513), 8, 65 ** 3)
    assert (nd, align) == (3, 8)
    b = slab.boundaries(513, 8, align)
    assert b == [0, 64, 128, 192, 256, 320, 384, 448, 513]
    # every boundary stays a whole row on every distributed level and each rank keeps >= 2 halos of rows
    for world in (2, 3, 4, 5, 8):
        for n in (33, 65, 129, 257, 513, 1025):
            nd, align = slab.plan_levels((n, n, n), world, 17 ** 3)
            b = slab.boundaries(n, world, align)
            assert b[0] == 0 and b[-1] == n and all(x % align == 0 for x in b[:-1])
            if nd:
                rows = min(y - x for x, y in zip(b, b[1:])) >> (nd - 1)
                assert rows >= 2 * slab.HALO
    assert slab.plan_levels((513, 513, 513), 1, 0)[0] == 0


CPU_CASES = [
    # world, shape (nz, ny, nx), dtype, omega, a1, a2, v, ncycles, zfull, crossover
    (2, (17, 65, 33), "f32", 1.0, 2, 2, 1, 2, True, 9 ** 3),
    (2, (33, 65, 17), "f64", 0.9, 3, 1, 2, 1, True, 9 ** 3),     # odd sweep counts, W-cycle, 3 levels split
    (3, (17, 129, 17), "f32", 1.0, 2, 2, 1, 2, False, 5 ** 3),   # uneven slabs, the reference's z range
    (2, (17, 33, 17), "f32", 0.8, 1, 1, 1, 2, True, 5 ** 3),
]


@pytest.mark.parametrize("case", CPU_CASES, ids=lambda c: "w%d-%s-%s" % (c[0], "x".join(map(str, c[1])), c[2]))
def test_slab_cycle_cpu_matches_whole_grid(case):
    world, shape, dtype, omega, a1, a2, v, ncycles, zfull, crossover = case
    launch(world, "cpu", shape, dtype, omega, a1, a2, v, ncycles, zfull, crossover)


GPU_CASES = [
    (2, (33, 65, 65), "f32", 1.0, 2, 2, 1, 2, True, 9 ** 3),
    (2, (65, 129, 33), "f64", 0.9, 3, 1, 2, 1, True, 17 ** 3),
    (3, (33, 129, 33), "f32", 1.0, 2, 2, 1, 2, False, 9 ** 3),
    # wide planes: the slabs of level 0 take the sweep that applies the correction itself
    (2, (9, 641, 513), "f32", 1.0, 2, 2, 1, 2, True, 5 * 321 * 257),
    (2, (9, 641, 513), "f32", 0.9, 1, 3, 2, 1, False, 5 * 321 * 257),
    # wide planes where no sweep carries the residual (double precision; no pre-smoothing): residual and
    # restriction as one march over the slab, the coarse rows at its ends from residual rows of their own
    (2, (9, 641, 513), "f64", 1.0, 2, 2, 1, 2, True, 5 * 321 * 257),
    (3, (9, 961, 321), "f64", 0.9, 2, 2, 2, 1, True, 5 * 481 * 161),
    (2, (9, 641, 513), "f32", 1.0, 0, 2, 1, 2, True, 5 * 321 * 257),
    # slabs whose planes have more tiles than the device runs at once: the fused sweeps go band by band
    (2, (9, 1281, 1025), "f64", 1.0, 2, 2, 1, 2, True, 5 * 641 * 513),
    (2, (9, 1281, 1025), "f32", 0.9, 2, 2, 1, 1, True, 5 * 641 * 513),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=lambda c: "w%d-%s-%s" % (c[0], "x".join(map(str, c[1])), c[2]))
def test_slab_cycle_gpu_matches_whole_grid(case):
    world, shape, dtype, omega, a1, a2, v, ncycles, zfull, crossover = case
    launch(world, "gpu", shape, dtype, omega, a1, a2, v, ncycles, zfull, crossover)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=lambda c: "w%d-%s-%s" % (c[0], "x".join(map(str, c[1])), c[2]))
def test_native_slab_plan_matches_whole_grid(case):
    """mgrt_slab_plan (the C++ driver inside libmgrt.so) with its callback transport."""
    world, shape, dtype, omega, a1, a2, v, ncycles, zfull, crossover = case
    launch(world, "native", shape, dtype, omega, a1, a2, v, ncycles, zfull, crossover)


FMG_CASES = [
    # ncycles is fmg's `iters` on the finest level (coarser levels run 3, as in the reference)
    (2, (33, 65, 65), "f32", 1.0, 2, 2, 1, 1, True, 9 ** 3),
    (2, (65, 129, 33), "f64", 0.9, 3, 1, 2, 2, True, 17 ** 3),
    (3, (33, 129, 33), "f32", 1.0, 2, 2, 1, 3, False, 9 ** 3),
    (2, (9, 641, 513), "f32", 1.0, 2, 2, 1, 1, True, 5 * 321 * 257),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", FMG_CASES, ids=lambda c: "w%d-%s-%s" % (c[0], "x".join(map(str, c[1])), c[2]))
def test_native_slab_plan_fmg_matches_whole_grid(case):
    """mgrt_slab_plan_fmg: every rank's rows bit for bit those of the oracle's whole-grid fmg."""
    world, shape, dtype, omega, a1, a2, v, iters, zfull, crossover = case
    launch(world, "native-fmg", shape, dtype, omega, a1, a2, v, iters, zfull, crossover)


@pytest.mark.gpu
@pytest.mark.parametrize("world, shape, dtype, faces", [
    (2, (33, 65, 65), "f32", (2, 2, 3, 3, 2, 3)),    # the faces the slabs are cut along are Neumann (frozen rows)
    (3, (17, 129, 33), "f64", (3, 2, 2, 3, 3, 2)),
    (2, (9, 641, 513), "f32", (2, 3, 3, 2, 2, 2)),   # wide slabs: the sweeps with residual and with correction
], ids=["w2-yN", "w3-mixed", "w2-wide"])
def test_native_slab_plan_with_neumann_faces(world, shape, dtype, faces):
    launch(world, "native", shape, dtype, 1.0, 2, 2, 1, 2, True, 9 ** 3 if shape[2] < 100 else 5 * 321 * 257, faces=faces)
    launch(world, "native-fmg", shape, dtype, 0.9, 2, 1, 1, 1, True, 9 ** 3 if shape[2] < 100 else 5 * 321 * 257, faces=faces)


@pytest.mark.gpu
def test_native_slab_plan_513_cubed_two_ranks():
    """The benchmark's grid cut in two: one V(2,2) cycle at 513^3, every rank's rows bit for bit those of
    the oracle's whole-grid cycle (two levels cut, 129^3 and below whole on both ranks)."""
    launch(2, "native", (513, 513, 513), "f32", 1.0, 2, 2, 1, 1, True, 129 ** 3, timeout=900)
