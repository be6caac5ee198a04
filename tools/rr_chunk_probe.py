#!/usr/bin/env python3
"""Residual + restriction of a latency-bound 3D level (mgrt_smooth_sweep_residual_restrict, iters == 0) against the
oracle's two kernels, bit for bit, and its time per call.  usage: rr_chunk_probe.py [n ...] [--f64] [--faces]"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from multigrid_amd import _lib as L  # noqa: E402
import _oracle  # noqa: E402


def main():
    f64 = "--f64" in sys.argv
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [257]
    dt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
    mgo = _oracle.load()
    L.init()
    dev = torch.device("cuda:0")
    for n in sizes:
        shape = (n, n, n) if n > 0 else (65, 257, 193)
        cs = tuple((s + 1) // 2 for s in shape)
        for faces in ((2,) * 6, (2, 3, 3, 2, 3, 2)):
            rng = np.random.default_rng(n)
            u = rng.standard_normal(shape).astype(dt)
            f = rng.standard_normal(shape).astype(dt)
            cell = mgo.box_cells(shape, faces)
            r = np.zeros_like(u)
            mgo.residual(cell, r, u, f)
            want = np.empty(cs, dt)
            mgo.restrict(cell, want, r)
            g, gc = L.grid(shape, faces=faces), L.grid(cs, faces=faces)
            du, df = torch.from_numpy(u).to(dev), torch.from_numpy(f).to(dev)
            got = torch.full(cs, 4.0, dtype=tdt, device=dev)
            work = torch.full(shape, 9.0, dtype=tdt, device=dev)
            L.smooth_sweep_residual_restrict(g, gc, None, got, du, df, 1.0, 0, r=work)
            torch.cuda.synchronize()
            h = got.cpu().numpy()
            bits = np.uint32 if dt is np.float32 else np.uint64
            bad = np.argwhere(h.view(bits) != want.view(bits))
            print(n, "f64" if f64 else "f32", faces, "equal" if len(bad) == 0 else "DIFFERS at %d points, first %s, planes %s" % (
                len(bad), bad[:4].tolist(), sorted(set(bad[:, 0].tolist()))[:20]), flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            L.smooth_sweep_residual_restrict(g, gc, None, got, du, df, 1.0, 0, r=work)
        reps = 50
        e0.record()
        for _ in range(reps):
            L.smooth_sweep_residual_restrict(g, gc, None, got, du, df, 1.0, 0, r=work)
        e1.record()
        torch.cuda.synchronize()
        print(n, "f64" if f64 else "f32", "%.1f us a call" % (e0.elapsed_time(e1) * 1e3 / reps), flush=True)


if __name__ == "__main__":
    main()
