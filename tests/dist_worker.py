"""One rank of a slab-decomposed V-cycle test.  Launched by tests/test_dist.py through
torch.distributed.run; exits non-zero when this rank's rows differ from the whole-grid oracle.

  dist_worker.py BACKEND NZ NY NX DTYPE OMEGA A1 A2 V NCYCLES ZFULL CROSSOVER [FACES]

FACES: six cell classes x-lo,x-hi,y-lo,y-hi,z-lo,z-hi (2 Dirichlet, 3 Neumann), default all Dirichlet.

BACKEND cpu: gloo, CPU tensors, oracle-backed kernels, the Python driver (multigrid_amd/dist.py).
BACKEND gpu: the same driver with the HIP kernels, every rank on cuda:0, gloo with host staging.
BACKEND native: mgrt_slab_plan of libmgrt.so (the C++ driver) on cuda:0 with callbacks over gloo.
BACKEND native-fmg: the same plan's full multigrid (NCYCLES = its iters) against the oracle's fmg.
BACKEND rccl, rccl-nograph: mgrt_slab_plan with the RCCL transport, every rank on the GPU of its LOCAL_RANK
(needs as many GPUs as ranks); halos, gathers and norms travel over RCCL, the cycle replayed from a graph
or launched directly."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    backend = sys.argv[1]
    nz, ny, nx = (int(a) for a in sys.argv[2:5])
    dt = np.float32 if sys.argv[5] == "f32" else np.float64
    omega, a1, a2, v, ncycles, zfull, crossover = (float(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]),
                                                   int(sys.argv[9]), int(sys.argv[10]), bool(int(sys.argv[11])),
                                                   int(sys.argv[12]))
    faces = tuple(int(c) for c in sys.argv[13].split(",")) if len(sys.argv) > 13 else (2,) * 6
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import _oracle
    from multigrid_amd import dist as slab
    o = _oracle.load("mgo")
    o.set_num_threads(1 if nz * ny * nx < 2e7 else 8)
    shape = (nz, ny, nx)
    rng = np.random.default_rng(1234)
    f = rng.standard_normal(shape).astype(dt)
    u0 = rng.standard_normal(shape).astype(dt)
    ref = u0.copy()
    hist_ref = []
    cell = o.box_cells(shape, faces)
    if faces != (2,) * 6:
        def rule(x, y, z, mx, my, mz):  # y faces win over x faces over z faces, on every level
            if y == 0: return faces[2]
            if y == my - 1: return faces[3]
            if x == 0: return faces[0]
            if x == mx - 1: return faces[1]
            if z == 0: return faces[4]
            if z == mz - 1: return faces[5]
            return 0
        o.set_celltype(rule)
        assert np.array_equal(o.level_cells(shape), cell), "the rule and box_cells disagree"

    def norm(x):
        r = np.zeros_like(x)
        o.residual(cell, r, x, f)
        return o.l2norm_exact(r)
    fmg = backend == "native-fmg"
    rccl = backend.startswith("rccl")
    use_graph = backend != "rccl-nograph"
    if rccl:
        backend = "native"
    if fmg:
        backend = "native"
        ref = o.fmg(f, dt(omega), a1, a2, v, ncycles, zfull)
    else:
        hist_ref.append(norm(ref))
        for _ in range(ncycles):
            o.iterate(ref, f, dt(omega), a1, a2, v, zfull)
            hist_ref.append(norm(ref))

    tdt = torch.float32 if dt == np.float32 else torch.float64
    if backend == "cpu":
        from _slab_backend import OracleKernels
        kernels, device = OracleKernels(), None
    else:
        from multigrid_amd import _lib as kernels
        local = int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0
        torch.cuda.set_device(local)
        kernels.init(local)
        device = torch.device("cuda", local)
    to = (lambda a: torch.from_numpy(a.copy())) if device is None else (lambda a: torch.from_numpy(a.copy()).to(device))
    if rccl:
        box = [kernels.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        cyc = kernels.SlabPlan(shape, tdt, transport="rccl", rank=rank, nranks=world, rccl_id=box[0], zfull=zfull,
                               crossover=crossover, faces=faces, use_graph=use_graph)
        ndist = cyc.distributed_levels
        lo, hi, a, b = cyc.rows()
    elif backend == "native":
        cyc = kernels.SlabPlan(shape, tdt, transport=kernels.GlooTransport(), zfull=zfull, crossover=crossover, faces=faces)
        ndist = cyc.distributed_levels
        lo, hi, a, b = cyc.rows()
    else:
        cyc = slab.SlabCycle(shape, tdt, kernels=kernels, device=device, zfull=zfull, crossover=crossover,
                             faces=faces if faces != (2,) * 6 else None)
        ndist = cyc.ndist
        lo, hi, a, b = cyc.local_rows()
    assert ndist >= 1, "this case does not split: pick a larger grid or a smaller crossover"
    ul, fl = cyc.scatter(to(u0)), cyc.scatter(to(f))
    if fmg:
        ul.fill_(float("nan"))  # the result does not depend on what the array held
        cyc.fmg(ul, fl, omega, a1, a2, v, ncycles)
        hist = []
    else:
        hist = cyc.vcycle(ul, fl, omega, a1, a2, v, ncycles, history=True)
    if device is not None:
        torch.cuda.synchronize()
    mine = cyc.owned(ul).cpu().numpy()
    want = ref[:, a:b, :]
    ok = np.array_equal(mine.view(np.uint8), want.view(np.uint8))
    hok = all(abs(h - r) <= 1e-9 * max(r, 1e-300) for h, r in zip(hist, hist_ref)) and len(hist) == len(hist_ref)
    if not ok or not hok:
        bad = np.argwhere(mine != want)
        print("rank %d/%d rows [%d,%d) ndist %d: mismatch at %d cells, first %s; hist %s vs %s"
              % (rank, world, a, b, ndist, len(bad), bad[:3].tolist(), hist, hist_ref), flush=True)
    flag = torch.tensor([0 if ok and hok else 1])
    dist.all_reduce(flag)
    cyc.close()
    dist.destroy_process_group()
    sys.exit(1 if flag.item() else 0)


if __name__ == "__main__":
    main()
