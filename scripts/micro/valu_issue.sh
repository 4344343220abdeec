#!/bin/bash
# usage (GPU box): scripts/micro/valu_issue.sh <tag>   -- builds and runs valu_issue.hip (timing pass), then a --pmc pass for
# SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_BUSY_CYCLES per kernel, and writes gpurun_out/valu_issue_<tag>.{log,json}
TAG=${1:-r05}; FROM=${2:-0}; TO=${3:-1000}   # FROM: first class index to run (v_fma_f64 always runs as the yardstick)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $REPO
OUT=$REPO/gpurun_out/valu_issue_$TAG; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o $OUT/valu_issue scripts/micro/valu_issue.hip || exit 1
$OUT/valu_issue 2000 full $FROM $TO > $OUT/timing.log || { tail -3 $OUT/timing.log; exit 1; }
cat $OUT/timing.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- $OUT/valu_issue 500 quick $FROM $TO > $OUT/pmc_run.log 2> $OUT/pmc_run.err || { tail -5 $OUT/pmc_run.err; exit 1; }
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, re, sys, collections
out, tag = sys.argv[1], sys.argv[2]
names = {}
rows = []  # timing
for line in open(out + "/timing.log"):
    m = re.match(r"^(.{44}) +(\d+) +([\d.]+) +([\d.]+) +([\d.]+) +([\d.]+)\s*$", line)
    if m: rows.append({"class": m.group(1).strip(), "waves_per_simd": int(m.group(2)), "ms": float(m.group(3)), "ns_per_inst_per_simd": float(m.group(4)),
                       "cycles_per_inst_per_simd_s_memtime": float(m.group(5)), "rel_to_fma_f64": float(m.group(6))})
classes = []
for r in rows:
    if r["class"] not in classes: classes.append(r["class"])
# pmc: dispatches come in launch order: class-major, W-minor (1 launch each in quick mode)
disp = collections.defaultdict(dict)
for f in glob.glob(out + "/pmc/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "valu_loop" in row["Kernel_Name"]:
            disp[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
            disp[int(row["Dispatch_Id"])]["kernel"] = row["Kernel_Name"]
ids = sorted(disp)
ws = sorted({r["waves_per_simd"] for r in rows})
print("\nPMC pass (500 trips): quad-cycles the VALU was held per instruction = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU; busy = ACTIVE*4 / (BUSY_CYCLES/32 * 1024)")
pmc = []
for i, d in enumerate(ids):
    c = disp[d]; cls = classes[i // len(ws)] if i // len(ws) < len(classes) else "?"; w = ws[i % len(ws)]
    q = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
    busy = c["SQ_ACTIVE_INST_VALU"] * 4 / (c["SQ_BUSY_CYCLES"] / 32 * 1024) if c.get("SQ_BUSY_CYCLES") else float("nan")
    pmc.append({"class": cls, "waves_per_simd": w, "kernel": c["kernel"], "active_per_inst_quadcycles": q, "valu_busy": busy,
                "insts_valu": c["SQ_INSTS_VALU"], "active_inst_valu": c["SQ_ACTIVE_INST_VALU"], "busy_cycles": c.get("SQ_BUSY_CYCLES"), "wave_cycles": c.get("SQ_WAVE_CYCLES")})
    print("%-44s W %d  ACTIVE/INSTS %.3f  busy %.3f" % (cls, w, q, busy))
json.dump({"tag": tag, "timing": rows, "pmc": pmc}, open(out + "/summary.json", "w"), indent=1)
PY
