#!/usr/bin/env python3
"""Per-level, per-kernel achieved bandwidth of the 3D V-cycle from a rocprofv3 --kernel-trace run of
bench.py.  The level of a dispatch follows from its place in the cycle: the level goes up by one
after every restriction, the one-launch bottom stands for all remaining levels, and every
interpolation belongs to the level above the one just left.

usage: level_report.py <rocprof output dir> [n=513] [elem_bytes=4]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

PEAK = 8000.0  # GB/s


def fused_args(name):
    m = re.search(r"k_smooth3d_fused<([^>]*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else None


def merge_bands(rows, sizes):
    """A fused sweep over a plane with more tiles than the device runs at once is issued as several launches of
    the same kernel, a band of tile rows each: make one row of them (first start to last end).  The tile shape
    follows from the template arguments <R, ITERS, SR, NW, PD, OMEGA1, RES, COR, STAGES, MAP, RST>; a run of
    equal launches is complete when its blocks add up to the tiles of a level."""
    out, run = [], None
    for t0, t1, name, blocks in rows:
        a = fused_args(name)
        if not a:
            out.append((t0, t1, name))
            run = None
            continue
        if run and run[2] == name:
            run[1], run[3] = t1, run[3] + blocks
            out[-1] = (run[0], run[1], name)
        else:
            run = [t0, t1, name, blocks]
            out.append((t0, t1, name))
        iters, sr, nw, res, stages = int(a[1]), int(a[2]), int(a[3]), a[6] == "true", int(a[8]) if len(a) > 8 else 1
        rst = len(a) > 10 and a[10] == "true"
        h = 2 * iters * stages + (1 if res else 0)
        ox, oy = 64 - 2 * (h + (1 if rst else 0)), nw * sr - 2 * h
        if any(run[3] == (-(-n // ox)) * (-(-n // oy)) for n in sizes):
            run = None  # complete: an equal launch that follows is an operation of its own
    return out


def main():
    root = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 513
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    sizes = [n]
    while sizes[-1] > 3:
        sizes.append((sizes[-1] + 1) // 2)
    pts = [s ** 3 for s in sizes]
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    # bench.py times the dominant kernel alone before the cycles; the cycles start with the residual norm of the
    # history (the first residual kernel that also sums)
    first = next((i for i, r in enumerate(rows) if re.search(r"k_residual3d_v<[^>]*true>", r[2])), 0)
    rows = merge_bands(rows[first:], sizes)
    # The last cycle ends where its way up from the last bottom launch is back on level 0: count the
    # restrictions before that launch down and the interpolations after it up.  What bench.py launches
    # after that (its timing loop over the dominant kernel) is not part of a cycle.
    def corrects(name):
        m = re.search(r"k_smooth3d_fused<([^>]*)>", name)
        return bool(m) and m.group(1).split(",")[7].strip() == "true"  # <R, ITERS, SR, NW, PD, OMEGA1, RES, COR, STAGES, MAP>
    bottoms = [i for i, r in enumerate(rows) if "k_bottom_cycle" in r[2]]
    if bottoms:
        # V-cycles: every cycle restricts the same number of times before its one bottom launch
        depth = sum("k_restrict" in r[2] or "k_residual_restrict3d" in r[2] for r in rows[:bottoms[-1]]) // len(bottoms)
        end = bottoms[-1]
        while depth > 0 and end + 1 < len(rows):
            end += 1
            if "k_interp3d_v" in rows[end][2] or corrects(rows[end][2]):
                depth -= 1
        if not corrects(rows[end][2]):  # the post-smoothing sweep after a correction of its own
            end = next((j for j in range(end + 1, len(rows)) if "k_smooth3d_fused" in rows[j][2]), len(rows) - 1)
        rows = rows[:end + 1]
    agg = defaultdict(list)
    level = 0
    seam_next = False
    for t0, t1, name in rows:
        m = re.search(r"(k_\w+)<([^>]*)>", name)
        if not m:
            continue
        k, targs = m.group(1), [a.strip() for a in m.group(2).split(",")]
        dur = (t1 - t0) * 1e-3
        if k == "k_smooth3d_fused_pair":  # two stages of one iteration each, two planes a step
            fam, byts = "sweep x2", 3 * pts[level] * w
        elif k == "k_smooth3d_fused":
            res, cor = targs[6] == "true", targs[7] == "true"
            if cor:
                level -= 1
            rst = len(targs) > 10 and targs[10] == "true"  # <R, ITERS, SR, NW, PD, OMEGA1, RES, COR, STAGES, MAP, RST>
            iters = int(targs[1]) * (int(targs[8]) if len(targs) > 8 else 1)  # iterations per stage x stages
            fam = "%ssweep x%d%s" % ("correct + " if cor else "", iters, " + residual + restrict" if rst else " + residual" if res else "")
            byts = (3 if rst else 4 if res else 3) * pts[level] * w + (pts[level + 1] * w if cor or rst else 0)
            seam_next = rst
        elif k == "k_residual3d_v":
            if targs[-1] == "true":     # with the norm: the history between cycles, level 0
                agg[(0, "residual + norm (history)", 3 * pts[0] * w)].append(dur)
                level = 0
                continue
            fam, byts = "residual", 3 * pts[level] * w
        elif k == "k_residual_restrict3d":
            fam, byts = "residual + restrict (tiles)", (2 * pts[level] + pts[level + 1]) * w
        elif k == "k_restrict" and seam_next:
            # the coarse rows on the seams between the sweep's rows of tiles (22 fine rows each): three fine
            # rows read and one coarse row written per seam
            seams = (sizes[level] + 21) // 22 - 1
            fam, byts = "restrict (seam rows)", seams * (3 * sizes[level] ** 2 + sizes[level + 1] ** 2) * w
        elif k == "k_restrict":
            fam, byts = "restrict", (pts[level] + pts[level + 1]) * w
        elif k == "k_interp3d_v":
            level -= 1
            fam, byts = "correct", (2 * pts[level] + pts[level + 1]) * w
        elif k == "k_bottom_cycle":
            fam, byts = "bottom cycle (levels %d..%d)" % (level, len(sizes) - 1), 0
        else:
            continue
        agg[(level, fam, byts)].append(dur)
        if k != "k_smooth3d_fused":
            seam_next = False
        if k in ("k_restrict", "k_residual_restrict3d"):
            level += 1
    print("3D %d^3, %d-byte elements; achieved = algorithmic bytes / mean kernel time; peak %.0f GB/s" % (n, w, PEAK))
    print("%-5s %-8s %-34s %6s %10s %10s %7s" % ("level", "extent", "kernel", "calls", "mean us", "GB/s", "% peak"))
    for (lv, fam, byts), v in sorted(agg.items()):
        mean = sum(v) / len(v)
        gbs = byts / mean / 1e3 if byts else 0.0
        print("%-5d %-8s %-34s %6d %10.1f %10.0f %6.1f%%" % (lv, "%d^3" % sizes[lv], fam, len(v), mean, gbs, 100 * gbs / PEAK))
    per_level = defaultdict(float)
    cycles = max(1, max(len(v) for (lv, fam, _), v in agg.items() if lv == 0 and "restrict" in fam and "history" not in fam))
    for (lv, fam, byts), v in agg.items():
        if "history" not in fam:
            per_level[lv] += sum(v) / cycles
    print("per cycle: " + ", ".join("L%d %.0f us" % (l, t) for l, t in sorted(per_level.items())) +
          "; total %.0f us" % sum(per_level.values()))


if __name__ == "__main__":
    main()
