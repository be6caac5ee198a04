#!/usr/bin/env python3
"""Turn separate FETCH_SIZE and WRITE_SIZE rocprofv3 passes (tools/pmc.sh <dir> fetch,write) into
HBM bytes per launch per kernel and write profiles/pmc_traffic.json, which bench.py reads.

Units and corrections for gfx950: rocprofv3 reports FETCH_SIZE and WRITE_SIZE in kilobytes
(1 KB = 1024 B); FETCH_SIZE tallies 128-byte read requests as 64 bytes, so streaming reads are
doubled; WRITE_SIZE is exact.

usage: pmc_traffic.py <pmc dir> <n> <dtype f32|f64> [out json]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(root, counter):
    vals = defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                m = re.search(r"(k_\w+)<([^>]*)>", row["Kernel_Name"])
                if m:
                    vals[(m.group(1), m.group(2))].append(float(row["Counter_Value"]))
    return {k: sum(v[1:] or v) / len(v[1:] or v) for k, v in vals.items()}


def main():
    root, n, dtype = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
    fetch = per_kernel(os.path.join(root, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(root, "write"), "WRITE_SIZE")
    rows = []
    for key in sorted(set(fetch) & set(write)):
        fb, wb = fetch[key] * 1024.0 * 2.0, write[key] * 1024.0
        rows.append({"kernel": key[0], "template": key[1], "n": n, "dtype": dtype,
                     "fetch_bytes": fb, "write_bytes": wb, "traffic_bytes": fb + wb,
                     "note": "FETCH_SIZE KB x 1024 x 2 (gfx950 half-count) + WRITE_SIZE KB x 1024, mean per launch"})
    try:
        with open(out) as fh:
            keys = {(r["kernel"], r["template"]) for r in rows}
            old = [r for r in json.load(fh) if not (r["n"] == n and r["dtype"] == dtype and (r["kernel"], r["template"]) in keys)]
    except (OSError, ValueError):
        old = []
    with open(out, "w") as fh:
        json.dump(old + rows, fh, indent=1)
    for r in rows:
        print("%-28s %-40s fetch %.1f MB  write %.1f MB" % (r["kernel"], r["template"][:40], r["fetch_bytes"] / 1e6, r["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
