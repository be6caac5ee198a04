#!/usr/bin/env python3
"""Headline benchmark: 3D 513^3 Poisson, Dirichlet box, float, V(2,2) multigrid cycles on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one V(2,2) cycle (omega = 1, pre/post smoothing 2+2, one coarse solve per level, smoother
over every z plane) on fields already resident in HBM.  Prints ONE JSON line:
value = finest-grid points x cycles / second over all GPUs.

For N > 1 the driver launches one process per GPU with torch.distributed.run; the 513^3 grid is cut
into N slabs along y (multigrid_amd/dist.py): halo exchange over RCCL before each fused sweep, the
levels of at most 129^3 points solved whole on every rank.  Total work is fixed: strong scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def make_problem(n, device):
    """tridimF1/tridimS1 of the reference (programstate.cpp:114-122) sampled as discretize_func does
    (functionhandler.cpp:64-84): interior F*h*h in float, Dirichlet faces the exact solution."""
    import math
    import torch
    c = (torch.arange(n, device=device, dtype=torch.float32) / (n - 1)).double()
    s = torch.sin(math.pi * c)
    sol = (s[:, None, None] * s[None, :, None] * s[None, None, :])
    dh = torch.tensor(1.0 / (n - 1), dtype=torch.float32, device=device)
    f = ((-3.0 * math.pi * math.pi) * sol).float() * dh * dh
    sol = sol.float()
    f[0], f[-1], f[:, 0], f[:, -1], f[:, :, 0], f[:, :, -1] = sol[0], sol[-1], sol[:, 0], sol[:, -1], sol[:, :, 0], sol[:, :, -1]
    return f.contiguous(), sol.contiguous()


def pmc_traffic(kernel, n, fields):
    """HBM bytes per launch of `kernel` at n^3 from the committed counter passes (profiles/
    pmc_traffic.json, written by tools/pmc_traffic.py from separate FETCH_SIZE and WRITE_SIZE
    rocprofv3 runs).  `fields` maps positions in the kernel's template argument list to the values
    that pick the variant timed here.  None when no pass for this size is on file: counters cannot
    be read from inside the timed process."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            for row in json.load(fh):
                args = [a.strip() for a in row["template"].split(",")]
                if row["kernel"] == kernel and row["n"] == n and row["dtype"] == "f32" and \
                        all(i < len(args) and args[i] == v for i, v in fields.items()):
                    return row["traffic_bytes"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(n_sample):
    """The oracle's C restatement (oracle/libmg_oracle.so, OpenMP) timed on this host on a bounded
    sample: one V(2,2) cycle of the same problem at n_sample^3."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    o = _oracle.load("mgo")
    if o is None:
        return None
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    o.set_num_threads(threads)
    shape = (n_sample,) * 3
    f, _ = o.discretize(0, shape, np.float32)
    u = np.zeros(shape, np.float32)
    o.mg(u, f, np.float32(1.0), 2, 2, 1, 1, True)  # warm: page in, spin threads up
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 10.0:
        o.mg(u, f, np.float32(1.0), 2, 2, 1, 1, True)
        reps += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = (time.perf_counter() - t0) / reps
    return {"value": n_sample ** 3 / dt, "unit": "DOF/s", "cores": threads, "kind": "port",
            "sample": "%d V(2,2) cycles of the same problem at %d^3 (float, full-z), %.3f s each"
                      % (reps, n_sample, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=513, help="finest extent per axis (2^k+1)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-n", type=int, default=257)
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a one-GPU box: every rank on cuda:0, gloo with host staging; checks that the "
                         "path runs, the number is not a measurement")
    ap.add_argument("--crossover", type=int, default=129 ** 3,
                    help="N > 1: levels with at most this many points are solved whole on every rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from multigrid_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if args.rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    _lib.init(local)

    n = args.n
    shape = (n, n, n)
    f, _ = make_problem(n, dev)
    u = torch.zeros_like(f)
    stream = torch.cuda.Stream()
    sync = torch.zeros(1, device=dev)
    torch.cuda.synchronize()  # the fields were produced on the default stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.all_reduce(sync.cpu() if args.rehearse else sync)
            torch.cuda.synchronize()

    t_sweep = None
    if world > 1:
        from multigrid_amd import dist as slab
        cyc = slab.SlabCycle(shape, torch.float32, device=dev, zfull=True, crossover=args.crossover,
                             use_graph=not args.no_graph)
        fl, ul = cyc.scatter(f), cyc.scatter(u)
        del f, u
        r0 = cyc.residual_norm(ul, fl)
        cyc.vcycle(ul, fl, 1.0, 2, 2, 1, args.warmup)
        barrier()
        t0 = time.perf_counter()
        cyc.vcycle(ul, fl, 1.0, 2, 2, 1, args.steps)
        barrier()
        dt = time.perf_counter() - t0
        r1 = cyc.residual_norm(ul, fl)
        layout = {"slab_rows": [b - a for a, b in zip(cyc.bounds, cyc.bounds[1:])], "halo_rows": slab.HALO,
                  "distributed_levels": cyc.ndist, "replicated_from": "x".join(map(str, cyc.coarse_shape))}
        cyc.close()

    with torch.cuda.stream(stream):
        if world == 1:
            plan = _lib.Plan(shape, torch.float32, zfull=True, use_graph=not args.no_graph)
            r0 = plan.vcycle(u, f, ncycles=0, history=True)[0]
            plan.vcycle(u, f, 1.0, 2, 2, 1, args.warmup)
            barrier()
            t0 = time.perf_counter()
            plan.vcycle(u, f, 1.0, 2, 2, 1, args.steps)
            barrier()
            dt = time.perf_counter() - t0
            r1 = plan.vcycle(u, f, ncycles=0, history=True)[0]
            plan.close()
            layout = None
        else:
            f, _ = make_problem(n, dev)
            u = torch.zeros_like(f)

        # dominant kernel: the pre-smoothing sweep of the finest level, which performs two red-black
        # iterations and leaves the residual behind (the longest launch of the cycle), timed alone
        # with events on the stream it runs on
        g = _lib.grid(shape)
        reps = 20
        work, resid = torch.empty_like(u), torch.empty_like(u)
        _lib.smooth_sweep_residual(g, work, resid, u, f, 1.0, 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps // 2):
            _lib.smooth_sweep_residual(g, work, resid, u, f, 1.0, 2)
            _lib.smooth_sweep_residual(g, u, resid, work, f, 1.0, 2)
        e1.record(stream)
        e1.synchronize()
        t_sweep = e0.elapsed_time(e1) * 1e-3 / reps

    if world > 1:
        t = torch.tensor([dt], device="cpu" if args.rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        dof = float(n) ** 3
        alg_bytes = 4.0 * dof * 4  # read u, read f, write the smoothed field, write the residual, once each
        achieved = alg_bytes / t_sweep / 1e9
        # <R, ITERS, SR, NW, PD, OMEGA1, RES, COR>: two iterations, omega == 1, with the residual pass
        traffic = pmc_traffic("k_smooth3d_fused", n, {0: "float", 1: "2", 5: "true", 6: "true", 7: "false"})
        # algorithmic bytes of a whole unfused V(2,2) cycle: 22 N w + 3 Nc w per level, 16 N w coarsest
        sizes, m = [], n
        while True:
            sizes.append(m)
            if m <= 3:
                break
            m = (m + 1) // 2
        cyc_bytes = sum((22 * s ** 3 + 3 * ((s + 1) // 2) ** 3) * 4 for s in sizes[:-1]) + 16 * sizes[-1] ** 3 * 4
        out = {
            "metric": "3D 513^3 V-cycle DOF/s", "value": dof * args.steps / dt, "unit": "DOF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"rehearsal": "all ranks on one GPU over gloo: not a measurement"} if args.rehearse else {}),
            "config": {"workload": "3D Poisson %d^3 Dirichlet box, V(2,2), omega=1, smoother over all z planes" % n,
                       "levels": len(sizes), "graph": not args.no_graph,
                       "parallelism": "single GPU" if world == 1 else "%d y-slabs, RCCL halo exchange" % world,
                       "layout": layout,
                       "residual_before": r0, "residual_after_warmup_and_steps": r1},
            "vcycles_per_s": args.steps / dt,
            "cycle_gbs_unfused_model": cyc_bytes * args.steps / dt / 1e9,
            "roofline": {"bound": "hbm", "kernel": "k_smooth3d_fused with the residual pass, 2 red-black iterations per sweep, finest level",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_s": t_sweep},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
