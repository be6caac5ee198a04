#!/usr/bin/env python3
"""What one GPU of an N-GPU run has to do, timed on a single GPU: the share of a middle rank of an
N-way y-slab split (mgrt_slab_plan with the stand-alone transport: packing, unpacking and copies of
the same size as the halos stay on the device, nothing crosses a wire).  T_N below is therefore a
lower bound of the N-GPU cycle time, and T_1 / (N * T_N) an upper bound of the parallel efficiency.

--weak times the other question: a grid that grows with the ranks (n x (512 N + 1) x n, every rank
keeps about 512 rows), where the share of a rank stays a full one-GPU problem and only the exchanges and
the gathered coarse levels come on top.

usage: tools/slab_proxy.py [--n 513] [--f64] [--crossover 4276737] [--cycles 20] [--weak]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=513)
    ap.add_argument("--f64", action="store_true")
    ap.add_argument("--crossover", type=int, default=129 * 129 * 257)
    ap.add_argument("--cycles", type=int, default=20)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--weak", action="store_true", help="the grid grows with the ranks: n x (512 N + 1) x n")
    args = ap.parse_args()
    import torch
    from multigrid_amd import _lib as L
    torch.cuda.set_device(0)
    L.init(0)
    dt = torch.float64 if args.f64 else torch.float32
    n = args.n
    rows = []
    t1 = None
    for world in [int(x) for x in args.ranks.split(",")]:
        shape = (n, 512 * world + 1, n) if args.weak else (n, n, n)
        for graph in (True, False):
            plan = L.SlabPlan(shape, dt, transport=None, rank=world // 2, nranks=world, crossover=args.crossover,
                              use_graph=graph)
            lo, hi, a, b = plan.rows()
            local = (n, hi - lo, n)
            # a smooth right-hand side keeps the iterates finite although the halos are not the neighbours'
            f = torch.zeros(local, device="cuda", dtype=dt)
            u = torch.zeros_like(f)
            plan.vcycle(u, f, 1.0, 2, 2, 1, 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.vcycle(u, f, 1.0, 2, 2, 1, args.cycles)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / args.cycles
            nd = plan.distributed_levels
            plan.close()
            del f, u
            if world == 1 and graph:
                t1 = t
            rows.append({"ranks": world, "graph": graph, "rows_owned": b - a, "rows_stored": hi - lo, "levels_cut": nd,
                         "ms_per_cycle": round(t * 1e3, 4),
                         "speedup_bound": round(t1 / t * (world if args.weak else 1), 3) if t1 else None,
                         "efficiency_bound": round(t1 / t / (1 if args.weak else world), 3) if t1 else None})
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
