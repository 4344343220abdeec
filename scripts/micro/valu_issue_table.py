#!/usr/bin/env python3
"""Merge the gpurun_out/valu_issue_<tag>/ runs of scripts/micro/valu_issue.sh into ONE table in shader CYCLES per
wave-instruction per SIMD -- SQ_BUSY_CYCLES / 32 (the kernel's span in cycles, whatever the clock did) over the instructions one
SIMD issued -- beside the wall-clock ns of the timing pass.  usage: valu_issue_table.py out.json tag[:from[:to]] ..."""
import csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WS = [1, 2, 3, 5, 7]
PMC_TRIPS, TIMING_TRIPS = 500, 2000
names = None
src = open(os.path.join(ROOT, "scripts", "micro", "valu_issue.hip")).read()
m = re.search(r"cls_name\[NCLS\] = \{(.*?)\};", src, re.S)
names = re.findall(r'"((?:[^"\\]|\\.)*)"', m.group(1))
rows = {}
for spec in sys.argv[2:]:
    parts = spec.split(":")
    tag, lo, hi = parts[0], int(parts[1]) if len(parts) > 1 else 0, int(parts[2]) if len(parts) > 2 else len(names)
    d = os.path.join(ROOT, "gpurun_out", f"valu_issue_{tag}")
    classes = [0] + [c for c in range(len(names)) if c != 0 and lo <= c < hi]
    # timing pass
    t = {}
    for line in open(os.path.join(d, "timing.log")):
        mm = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if mm and mm.group(1).strip() in names:
            t[(names.index(mm.group(1).strip()), int(mm.group(2)))] = float(mm.group(4))
    # pmc pass: dispatches in launch order = class-major, W-minor
    disp = {}
    for f in glob.glob(os.path.join(d, "pmc", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "valu_loop" in r["Kernel_Name"]:
                disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    assert len(ids) == len(classes) * len(WS), (tag, len(ids), len(classes))
    for i, did in enumerate(ids):
        c, w = classes[i // len(WS)], WS[i % len(WS)]
        if c == 0 and tag != sys.argv[2].split(":")[0]:
            continue  # the yardstick of a later run: kept from the first
        k = disp[did]
        insts_per_simd = w * PMC_TRIPS * 256.0
        rows[(c, w)] = {"class": names[c], "waves_per_simd": w,
                        "cycles_per_inst": round(k["SQ_BUSY_CYCLES"] / 32.0 / insts_per_simd, 3),
                        "ns_per_inst": t.get((c, w)),
                        "sq_active_inst_valu_per_inst": round(k["SQ_ACTIVE_INST_VALU"] / k["SQ_INSTS_VALU"], 3),
                        "run": tag}
out = [rows[k] for k in sorted(rows)]
json.dump({"what": "scripts/micro/valu_issue.hip on one MI355X: cycles = SQ_BUSY_CYCLES / 32 / (W x trips x 256 instructions); "
                   "ns from HIP events of a separate pass; sq_active_inst_valu_per_inst = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU "
                   "(the counter charges one quad-cycle per plain instruction whatever it really costs)",
           "rows": out}, open(sys.argv[1], "w"), indent=1)
print("%-48s %s" % ("class", "  ".join(f"W={w}" for w in WS)))
for c in range(len(names)):
    if (c, 1) in rows:
        print("%-48s %s" % (names[c], "  ".join("%5.2f" % rows[(c, w)]["cycles_per_inst"] for w in WS)))
