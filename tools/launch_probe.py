#!/usr/bin/env python3
"""What a call of mgrt_plan_vcycle costs from the host's point of view: wall time from the call to the end of
the device work, for 1, 5 and 20 cycles, replayed from the captured graph and launched directly.  The
difference between n cycles in one call and n times the steady cycle time is what starting the first cycle
costs.

usage: launch_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multigrid_amd import _lib


def main():
    _lib.init(0)
    dev = torch.device("cuda", 0)
    cases = [((513, 513, 513), torch.float32), ((257, 257, 257), torch.float32), ((65, 65, 65), torch.float32),
             ((257, 257), torch.float32), ((4097, 4097), torch.float32)]
    print("%-14s %-7s %8s %12s %12s" % ("grid", "launch", "cycles", "us / call", "us / cycle"))
    for shape, dt in cases:
        g = torch.Generator(device="cpu").manual_seed(3)
        f = torch.randn(shape, generator=g, dtype=dt).to(dev)
        for graph in (True, False):
            plan = _lib.Plan(shape, dt, zfull=True, use_graph=graph)
            u = torch.zeros(shape, dtype=dt, device=dev)
            plan.vcycle(u, f, 1.0, 2, 2, 1, 5)
            torch.cuda.synchronize()
            for n in (1, 5, 20):
                best = 1e9
                for _ in range(7):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    plan.vcycle(u, f, 1.0, 2, 2, 1, n)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                print("%-14s %-7s %8d %12.1f %12.1f" % ("x".join(map(str, shape)), "graph" if graph else "direct", n,
                                                       best * 1e6, best * 1e6 / n))
            plan.close()


if __name__ == "__main__":
    main()
