#!/usr/bin/env python3
"""development: time 2D V(2,2) cycles (4097^2, x faces Dirichlet, y faces Neumann)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import _lib as L
L.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
shape = (n, n)
faces = (L.DIRICHLET, L.DIRICHLET, L.NEUMANN, L.NEUMANN, L.DIRICHLET, L.DIRICHLET)
f = torch.randn(shape, device="cuda")
u = torch.zeros_like(f)
torch.cuda.synchronize()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    plan = L.Plan(shape, torch.float32, faces=faces, use_graph=True)
    plan.vcycle(u, f, 1.0, 2, 2, 1, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.vcycle(u, f, 1.0, 2, 2, 1, 20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("2D %d^2 V(2,2): %.3f ms per cycle, %.2f GDOF/s, unfused-model %.0f GB/s" % (n, dt * 1e3, n * n / dt / 1e9, 22.75 * n * n * 4 * 4 / 3 / dt / 1e9))
    plan.close()
