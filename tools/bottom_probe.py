#!/usr/bin/env python3
"""Where the one-launch bottom of a cycle spends its time: a plan whose finest level already fits the
bottom kernel is one launch per cycle, so cycles of it with and without smoothing iterations separate
the smoother's marches from the residual / restriction / correction phases.

usage: bottom_probe.py [--reps 400]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multigrid_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=400)
    args = ap.parse_args()
    _lib.init(0)
    dev = torch.device("cuda", 0)
    cases = [((17, 17, 17), torch.float32), ((17, 17, 17), torch.float64), ((9, 9, 9), torch.float32),
             ((65, 65), torch.float32), ((33, 33), torch.float32)]
    print("%-14s %-4s %-18s %10s" % ("finest level", "type", "a1,a2,coarse solves", "us / cycle"))
    for shape, dt in cases:
        g = torch.Generator(device="cpu").manual_seed(3)
        f = torch.randn(shape, generator=g, dtype=dt).to(dev)
        for a1, a2, v in ((2, 2, 1), (1, 1, 1), (0, 0, 1), (2, 2, 0)):
            plan = _lib.Plan(shape, dt, zfull=True, use_graph=True)
            u = torch.zeros(shape, dtype=dt, device=dev)
            plan.vcycle(u, f, 1.0, a1, a2, v, 3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.vcycle(u, f, 1.0, a1, a2, v, args.reps)
            e1.record()
            torch.cuda.synchronize()
            print("%-14s %-4s %-18s %10.2f" % ("x".join(map(str, shape)), "f32" if dt == torch.float32 else "f64",
                                               "%d,%d,%d" % (a1, a2, v), e0.elapsed_time(e1) * 1e3 / args.reps))
            plan.close()


if __name__ == "__main__":
    main()
