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
st = torch.cuda.Stream()
for name, shape, dt, faces in CONFIGS:
    f = torch.randn(shape, device="cuda", dtype=dt)
    u = torch.zeros_like(f)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        plan = L.Plan(shape, dt, faces=faces, zfull=True, use_graph=True)
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
