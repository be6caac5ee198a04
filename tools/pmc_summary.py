#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv files under a directory) per kernel:
mean counter value per dispatch, skipping the first (warm-up) dispatch of each kernel."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+)<([^>]*)>", name)
    return "%s<%s>" % (m.group(1), m.group(2)) if m else name[:60]


def main(root):
    vals = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                vals[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(vals):
        if not k.startswith("k_"):
            continue
        parts = []
        for c in sorted(vals[k]):
            v = vals[k][c][1:] or vals[k][c]
            parts.append("%s=%.4g" % (c, sum(v) / len(v)))
        print("%-48s %s" % (k, "  ".join(parts)))


if __name__ == "__main__":
    main(sys.argv[1])
