#!/usr/bin/env python3
"""Time to tolerance: V(2,2) cycles on the 3D N^3 Dirichlet problem until ||r||_2 / ||r_0||_2 <= tol.
float floors near 2e-6 (the residual is evaluated in float), so 1e-8 needs double."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import torch
from multigrid_amd import _lib as L
from bench import make_problem

L.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
for dt in (torch.float64, torch.float32):
    f = make_problem((n, n, n), dt, (2,) * 6, torch.device("cuda"))
    u = torch.zeros_like(f)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan = L.Plan((n, n, n), dt, zfull=True, use_graph=True)
        plan.vcycle(torch.zeros_like(f), f, 1.0, 2, 2, 1, 2)  # warm the graph on a scratch unknown
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = plan.vcycle(u, f, ncycles=0, history=True)
        r0, cycles = hist[0], 0
        rel = [1.0]
        while rel[-1] > tol and cycles < 40:
            h = plan.vcycle(u, f, 1.0, 2, 2, 1, 1, history=True)
            cycles += 1
            rel.append(h[-1] / r0)
            if len(rel) > 3 and rel[-1] > 0.9 * rel[-2]:
                break  # at the rounding floor
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        plan.close()
    print(json.dumps({"n": n, "dtype": str(dt).split(".")[-1], "tol": tol, "cycles": cycles, "reached": rel[-1] <= tol,
                      "relative_residual": rel[-1], "seconds_including_residual_checks": round(dt_s, 4),
                      "DOF_per_s_to_tolerance": round(n ** 3 / dt_s / 1e9, 3) if rel[-1] <= tol else None,
                      "contraction_per_cycle": round((rel[-1]) ** (1.0 / max(1, cycles)), 4)}), flush=True)
