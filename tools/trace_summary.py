#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name and grid size, calls, mean and total time."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+)<([^>]*)>", name)
    return "%s<%s>" % (m.group(1), m.group(2)) if m else re.sub(r"\(.*", "", name)[:50]


def main(root):
    agg = defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
                key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"]))
                agg[key].append(dur)
    tot = sum(sum(v) for v in agg.values())
    print("%-52s %10s %6s %10s %10s %6s" % ("kernel", "grid", "calls", "mean_us", "total_us", "%"))
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-52s %10d %6d %10.1f %10.1f %6.2f" % (k, g, len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))


if __name__ == "__main__":
    main(sys.argv[1])
