#!/usr/bin/env python3
"""Per-kernel timing at one level size (default 513^3, float): achieved GB/s from algorithmic bytes.
Development aid; bench.py is the measurement of record."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=513)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--f64", action="store_true")
    ap.add_argument("--only", default="", help="substring filter on kernel names")
    ap.add_argument("--dim", type=int, default=3, help="2: the 2D sweeps on an n x n grid")
    ap.add_argument("--one-grid", action="store_true", help="fused 3D sweeps: a large plane as one launch instead of band by band")
    args = ap.parse_args()
    import torch
    from multigrid_amd import _lib as L
    torch.cuda.set_device(0)
    L.init(0)
    if args.one_grid:
        L.sweep_parts(3 | 4)
    n = args.n
    dt = torch.float64 if args.f64 else torch.float32
    w = 8 if args.f64 else 4
    if args.dim == 2:
        return bench2d(args, torch, L, dt, w)
    shape = (n, n, n)
    nc = (n + 1) // 2
    N = float(n) ** 3
    Nc = float(nc) ** 3
    u = torch.randn(shape, device="cuda", dtype=dt)
    f = torch.randn(shape, device="cuda", dtype=dt)
    o = torch.empty_like(u)
    r = torch.empty_like(u)
    e = torch.randn((nc, nc, nc), device="cuda", dtype=dt)
    rc = torch.empty_like(e)
    g, gc = L.grid(shape), L.grid((nc, nc, nc))
    stream = torch.cuda.Stream()

    def timeit(name, fn, nbytes):
        if args.only and args.only not in name:
            return
        with torch.cuda.stream(stream):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.reps):
                fn()
            e1.record(stream)
            e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / args.reps
        print("%-36s %9.1f us  %8.1f GB/s  (%4.1f%% of 8 TB/s)" % (name, t * 1e6, nbytes / t / 1e9, nbytes / t / 8e10))

    timeit("copy (torch)", lambda: o.copy_(u), 2 * N * w)
    timeit("smooth_pass", lambda: L.smooth_pass(g, u, f, 1.0, 0), 2 * N * w)
    timeit("smooth_sweep x1", lambda: L.smooth_sweep(g, o, u, f, 1.0, 1), 3 * N * w)
    timeit("smooth_sweep x2", lambda: L.smooth_sweep(g, o, u, f, 1.0, 2), 3 * N * w)
    timeit("smooth_sweep x2 omega.8", lambda: L.smooth_sweep(g, o, u, f, 0.8, 2), 3 * N * w)
    timeit("smooth_sweep+res x2", lambda: L.smooth_sweep_residual(g, o, r, u, f, 1.0, 2), 4 * N * w)
    timeit("smooth_sweep+res x1", lambda: L.smooth_sweep_residual(g, o, r, u, f, 1.0, 1), 4 * N * w)
    if L.restrict_sweep_supported(g, dt, 2):
        timeit("sweep+res+restrict x2 (2 launches)", lambda: L.smooth_sweep_residual_restrict(g, gc, o, rc, u, f, 1.0, 2, r=r),
               (3 * N + Nc) * w)

        def sweep_alone():
            L.sweep_parts(1)
            L.smooth_sweep_residual_restrict(g, gc, o, rc, u, f, 1.0, 2, r=r)
            L.sweep_parts(3)
        timeit("sweep+res+restrict x2 (sweep)", sweep_alone, (3 * N + Nc) * w)
    timeit("residual", lambda: L.residual(g, r, u, f), 3 * N * w)
    n2 = torch.zeros(1, device="cuda", dtype=torch.float64)
    timeit("residual+norm", lambda: L.residual(g, r, u, f, norm2=n2), 3 * N * w)
    timeit("restrict", lambda: L.restrict(g, gc, rc, r), (N + Nc) * w)
    timeit("correct (in place)", lambda: L.correct(g, gc, u, u, e), (2 * N + Nc) * w)
    try:
        timeit("correct+smooth_sweep x2", lambda: L.correct_smooth_sweep(g, gc, o, u, e, f, 1.0, 2), (3 * N + Nc) * w)
    except L.MgrtError as err:
        print("correct+smooth_sweep x2: %s" % err)
    timeit("prolong", lambda: L.prolong(g, gc, o, e), (N + Nc) * w)
    timeit("sumsq", lambda: L.sumsq(g, u, n2), N * w)


def bench2d(args, torch, L, dt, w):
    n = args.n
    nc = (n + 1) // 2
    N, Nc = float(n) ** 2, float(nc) ** 2
    u = torch.randn((n, n), device="cuda", dtype=dt)
    f = torch.randn((n, n), device="cuda", dtype=dt)
    o = torch.empty_like(u)
    e = torch.randn((nc, nc), device="cuda", dtype=dt)
    rc = torch.empty_like(e)
    g, gc = L.grid((n, n)), L.grid((nc, nc))
    stream = torch.cuda.Stream()

    def timeit(name, fn, nbytes):
        with torch.cuda.stream(stream):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.reps):
                fn()
            e1.record(stream)
            e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / args.reps
        print("%-36s %9.1f us  %8.1f GB/s  (%4.1f%% of 8 TB/s)" % (name, t * 1e6, nbytes / t / 1e9, nbytes / t / 8e10))

    timeit("2D smooth_sweep x2", lambda: L.smooth_sweep(g, o, u, f, 1.0, 2), 3 * N * w)
    timeit("2D sweep+res+restrict x2", lambda: L.smooth_sweep_residual_restrict(g, gc, o, rc, u, f, 1.0, 2), (3 * N + Nc) * w)
    timeit("2D correct+sweep x2", lambda: L.correct_smooth_sweep(g, gc, o, u, e, f, 1.0, 2), (3 * N + Nc) * w)


if __name__ == "__main__":
    main()
