#!/usr/bin/env python3
"""Condense scripts/calib_gather.sh's rocprofv3 passes: per calibration kernel and counter the mean value per launch,
next to the bytes the kernel must move (lines x 128 B touched once; lanes x 16 B requested).
Usage: calib_gather_summary.py gpurun_out/calib_<tag>   -> <dir>/summary.json (+ stdout table)"""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
PIECES = {"g16": 1, "g32": 2, "g80": 5, "g128": 8}
LINES = {"small": 1 << 20, "large": 1 << 23}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if name.startswith("g") and name.split("_")[0] in PIECES:
            vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"note": "every kernel touches each 128-B line of its buffer once; FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 "
               "in units of 1024 B (converted to bytes here); TCC_* counters are raw event counts", "kernels": {}}
for name in sorted(vals):
    pat, buf, temp = name.split("_")
    lines = LINES[buf]
    k = {"lines": lines, "line_bytes_touched": lines * 128, "lane_bytes_requested": lines * 16 * PIECES[pat]}
    for c, v in sorted(vals[name].items()):
        m = sum(v) / len(v)
        k[c] = m * 1024.0 if c in ("FETCH_SIZE", "WRITE_SIZE") else m
    if "FETCH_SIZE" in k:
        k["FETCH_SIZE_per_line"] = k["FETCH_SIZE"] / lines
    out["kernels"][name] = k
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
cols = sorted({c for k in out["kernels"].values() for c in k if c not in ("lines", "line_bytes_touched", "lane_bytes_requested")})
print("kernel".ljust(18) + "".join(c[-22:].rjust(24) for c in cols))
for name, k in out["kernels"].items():
    print(name.ljust(18) + "".join((("%.4g" % k[c]) if c in k else "-").rjust(24) for c in cols))
