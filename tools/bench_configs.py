#!/usr/bin/env python3
"""Cycle time of every benchmark configuration that fits one GPU (bench.py reports only the headline
one): V(2,2), omega 1, fields resident in HBM, graph replay, 3D smoother over all planes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import _lib as L

L.init(0)
D, N = L.DIRICHLET, L.NEUMANN
CONFIGS = [
    ("2D 257^2 Dirichlet f32", (257, 257), torch.float32, (D,) * 6),
    ("2D 4097^2 x Dirichlet / y Neumann f32", (4097, 4097), torch.float32, (D, D, N, N, D, D)),
    ("3D 257^3 Neumann f32", (257,) * 3, torch.float32, (N,) * 6),
    ("3D 257^3 Dirichlet f32", (257,) * 3, torch.float32, (D,) * 6),
    ("3D 513^3 Dirichlet f32", (513,) * 3, torch.float32, (D,) * 6),
    ("3D 513^3 Dirichlet f64", (513,) * 3, torch.float64, (D,) * 6),
    ("3D 1025^3 Dirichlet f64", (1025,) * 3, torch.float64, (D,) * 6),
]
def ball_map(shape):
    """A domain that is not a box, as one byte per point: the ball inscribed in the grid (inner), a shell
    that is Dirichlet on the low-x side and Neumann on the other, outer cells around it; the faces of
    the array Dirichlet."""
    import numpy as np
    ax = [(np.arange(n) - 0.5 * (n - 1)) / (0.5 * (n - 1)) for n in shape]
    q = ax[0][:, None, None] ** 2 + ax[1][None, :, None] ** 2 + ax[2][None, None, :] ** 2
    c = np.full(shape, L.OUTER, np.uint8)
    c[q < 1.0] = L.NEUMANN
    c[(q < 1.0) & (ax[2][None, None, :] < 0)] = L.DIRICHLET
    c[q < 0.81] = L.INNER
    for d in range(3):
        idx = [slice(None)] * 3
        for e in (0, -1):
            idx[d] = e
            c[tuple(idx)] = L.DIRICHLET
    return c


CONFIGS.insert(4, ("3D 257^3 ball domain (byte map on every level) f32", (257,) * 3, torch.float32, "ball"))
st = torch.cuda.Stream()
for name, shape, dt, faces in CONFIGS:
    f = torch.randn(shape, device="cuda", dtype=dt)
    u = torch.zeros_like(f)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        plan = L.Plan(shape, dt, faces=(D,) * 6 if faces == "ball" else faces, zfull=True, use_graph=True)
        if faces == "ball":
            for l in range(plan.levels):
                plan.set_cells(l, ball_map(plan.level_shape(l)))
        plan.vcycle(u, f, 1.0, 2, 2, 1, 3)
        torch.cuda.synchronize()
        reps = 20 if shape[0] < 1000 or len(shape) == 2 else 5
        t0 = time.perf_counter()
        plan.vcycle(u, f, 1.0, 2, 2, 1, reps)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / reps
        plan.close()
    dof = 1
    for s in shape:
        dof *= s
    print(json.dumps({"config": name, "ms_per_cycle": round(dt_s * 1e3, 4), "GDOF_per_s": round(dof / dt_s / 1e9, 2)}),
          flush=True)
    del f, u
