#!/bin/bash
# usage (GPU box): scripts/kernel_clock.sh <tag> [bench args]  -- the shader clock DURING the render kernel: GRBM_GUI_ACTIVE (summed over the
# 8 XCDs) / 8 / the kernel's duration from the same pass's kernel trace is not available in one pass, so: cycles from --pmc, time from bench.py's
# own event timing of the same launch (kernel_ms in its line).
set -uo pipefail
TAG=${1:-r05}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$REPO" || exit 1
OUT="$REPO/gpurun_out/clock_$TAG"; mkdir -p "$OUT" || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/pmc" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rccl-probe "$@" > "$OUT/bench.json" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
line = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
ms = line["roofline"]["kernel_ms"]; name = line["roofline"]["kernel"]
rows = {}
for f in glob.glob(out + "/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if name + "<false" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
v = [d for d in rows.values() if "GRBM_GUI_ACTIVE" in d]
gui = sum(d["GRBM_GUI_ACTIVE"] for d in v) / len(v); busy = sum(d["SQ_BUSY_CYCLES"] for d in v) / len(v)
print(f"{name}: kernel {ms:.3f} ms (HIP events, profiled pass); GRBM_GUI_ACTIVE/8 = {gui/8:.4g} cycles -> {gui/8/(ms*1e-3)/1e9:.3f} GHz; "
      f"SQ_BUSY_CYCLES/32 = {busy/32:.4g} cycles -> {busy/32/(ms*1e-3)/1e9:.3f} GHz  ({len(v)} launches)")
PY
