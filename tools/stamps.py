#!/usr/bin/env python3
"""Where a wavefront of the fused 3D sweep spends a time step: runs one sweep on a diagnostic build of
the library (make stamps -> multigrid_amd/libmgrt_stamps.so, the kernel source with the stamp hooks of tools/diag/mgrt_stamps.hip) and prints,
for every wavefront of the middle block, shader-clock cycles per fast step and segment.
usage: tools/stamps.py [--n 257] [--iters 2] [--res]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MGRT_LIB"] = os.path.join(ROOT, "multigrid_amd", "libmgrt_stamps.so")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=257)
    ap.add_argument("--ny", type=int, default=0, help="rows, when the grid is a y-slab of n^3")
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--f64", action="store_true")
    args = ap.parse_args()
    import torch
    from multigrid_amd import _lib as L
    L.init(0)
    lib = L.load()
    dt = torch.float64 if args.f64 else torch.float32
    shape = (args.n, args.ny or args.n, args.n)
    u = torch.randn(shape, device="cuda", dtype=dt)
    f = torch.randn(shape, device="cuda", dtype=dt)
    o, r = torch.empty_like(u), torch.empty_like(u)
    g = L.grid(shape)
    for _ in range(3):
        if args.res:
            L.smooth_sweep_residual(g, o, r, u, f, 1.0, args.iters)
        else:
            L.smooth_sweep(g, o, u, f, 1.0, args.iters)
    buf = (C.c_ulonglong * 128)()
    lib.mgrt_debug_stamps.argtypes = [C.c_void_p]
    L.check(lib.mgrt_debug_stamps(buf))
    names = ["wait prefetch", "insert+publish", "barrier", "LDS+passes", "store/hand"]
    print("%dx%dx%d %s iters=%d res=%s: cycles per fast step (shader clock)" % (shape[2], shape[1], shape[0], "f64" if args.f64 else "f32", args.iters, args.res))
    print("wave " + " ".join("%15s" % n for n in names) + "      total  steps")
    for w in range(16):
        row = buf[8 * w:8 * w + 8]
        if not row[7]:
            continue
        per = [row[k] / row[7] for k in range(5)]
        print("%4d " % w + " ".join("%15.0f" % p for p in per) + " %10.0f %6d" % (sum(per), row[7]))


if __name__ == "__main__":
    main()
