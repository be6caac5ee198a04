#!/usr/bin/env python3
"""development: time the one-launch bottom cycle for a few smoothing counts and sizes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import _lib as L

L.init(0)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for n in (17, 9, 5):
        shape = (n, n, n)
        f = torch.randn(shape, device="cuda")
        u = torch.zeros_like(f)
        plan = L.Plan(shape, torch.float32, zfull=True, use_graph=False)
        for a1, a2, v in ((2, 2, 1), (1, 1, 1), (0, 0, 1), (2, 2, 0), (0, 0, 0)):
            plan.vcycle(u, f, 1.0, a1, a2, v, 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            plan.vcycle(u, f, 1.0, a1, a2, v, 50)
            e1.record(st)
            e1.synchronize()
            print("n=%d a1=%d a2=%d v=%d: %.1f us per cycle" % (n, a1, a2, v, e0.elapsed_time(e1) * 1e3 / 50))
        plan.close()
