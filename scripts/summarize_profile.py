#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ rocprofv3 run (scripts/profile_gpu.sh) into profiles/:
  profiles/<tag>_kernel_stats.csv   -- rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_pmc.json           -- per-launch counters of the render kernel + the bench lines
Usage: scripts/summarize_profile.py <tag> [fetch_scale]
fetch_scale: factor applied to FETCH_SIZE for the HBM-bytes figure (see DESIGN.md "Traffic counters").
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2 or sys.argv[1].startswith("-"):
    sys.exit(__doc__)   # (`--help` used to be taken for a tag and wrote profiles/--help_pmc.json)
tag = sys.argv[1]
if not os.path.isdir(os.path.join(ROOT, "gpurun_out", f"prof_{tag}")):
    sys.exit(f"gpurun_out/prof_{tag}/ not found (scripts/profile_gpu.sh {tag} writes it on the GPU box)")
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def _render_ns(path):
    """total duration of the render kernels in one rocprofv3 kernel_stats.csv"""
    try:
        return sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(path)) if "render_" in r["Name"])
    except Exception:
        return -1.0


# one file per traced process: bench.py itself and the world-size-1 RCCL child it starts after the timed region
# (bench.py rccl_probe, a tiny frame) -- the summary wanted is the bench's, the one that spent the most time in render kernels
stats_files = sorted(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=_render_ns)
if stats_files:
    shutil.copy(stats_files[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))

counters = collections.defaultdict(lambda: collections.defaultdict(list))
meta = collections.defaultdict(dict)
def _render_rows(path):
    try:
        return sum(1 for r in csv.DictReader(open(path)) if "render_" in r["Kernel_Name"])
    except Exception:
        return -1


for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    # one file per profiled process: keep the bench's own (the one with the most render-kernel rows), never a child's
    files = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=_render_rows)
    for f in files[-1:]:
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "render_" not in name:
                continue
            # pmc_sq2 repeats SQ_ACTIVE_INST_VALU next to SQ_THREAD_CYCLES_VALU: keep that pair apart (same pass)
            cname = row["Counter_Name"] + ("@2" if d.endswith("pmc_sq2") and row["Counter_Name"] == "SQ_ACTIVE_INST_VALU" else "")
            counters[name][cname].append(float(row["Counter_Value"]))
            for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                meta[name][k] = float(row[k])

# FETCH_SIZE calibration kernels (scripts/calib_fetch.hip): known 1 GiB per launch
calib = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_calib", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE" and row["Kernel_Name"].startswith("read"):
            calib[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]) * 1024.0)

bench = {}
for f in glob.glob(os.path.join(src, "bench_*.json")):
    try:
        bench[os.path.basename(f)[:-5]] = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        pass

out = {"tag": tag, "kernels": {}, "bench_lines": bench}
if calib:
    out["fetch_calibration"] = {k: {"fetch_size_bytes": v, "true_bytes": float(1 << 30),
                                    "true_over_reported": [float(1 << 30) / x if x else None for x in v]}
                                for k, v in calib.items()}
workload = scene = kname = spl = build = None
for b in bench.values():
    build = b.get("build_id", build)
    workload = b.get("config", {}).get("workload", workload)
    scene = b.get("config", {}).get("scene", scene)
    kname = b.get("roofline", {}).get("kernel", kname)
    spl = b.get("roofline", {}).get("samples_per_launch", spl)
out["workload"] = workload
out["scene"] = scene                 # bench.py matches a profile by scene + kernel and scales the counters per sample
out["kernel"] = kname
out["samples_per_launch"] = spl
# which binary this is a profile of: the library's own id (flux_build_id, printed by bench.py) and the checkout it was summarised in
out["build_id"] = build
try:
    import subprocess
    out["git_head"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    out["git_dirty"] = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "flux_amd/csrc", "include"], capture_output=True,
                                           text=True).stdout.strip())
except Exception:
    out["git_head"] = None
# dynamic VALU instruction classes of the same build (scripts/valu_classes.sh <tag> ...), when that pass was run: the mean issue
# cost of one VALU instruction follows from them and the per-class costs measured by scripts/micro/valu_issue.hip
# (profiles/r05_valu_issue.json): f64 add / mul / fma 4.13 cycles, f64 transcendental seeds 16.1, f32 fma (the packed filter) 4.13,
# plain f32 add / mul 2.45, conversions 4.67, int32 3.3, int64 4.13, and the instructions no class counts (moves, compares, selects,
# lane operations) 3.43 -- their static mix in round 5's listing (170 moves x 2.45, 100 compares x 4.75, 90 selects x 3.3, 42 others x 4.5)
vc = os.path.join(ROOT, "gpurun_out", f"valu_{tag}", "summary.json")
if os.path.exists(vc):
    per = json.load(open(vc))["per_64_samples"]
    cost = {"ADD_F64": 4.13, "MUL_F64": 4.13, "FMA_F64": 4.13, "TRANS_F64": 16.1, "ADD_F32": 2.45, "MUL_F32": 2.45, "FMA_F32": 4.13,
            "TRANS_F32": 4.13, "CVT": 4.67, "INT32": 3.3, "INT64": 4.13}
    tot = per["SQ_INSTS_VALU"]
    cl = {c: per.get("SQ_INSTS_VALU_" + c, 0.0) for c in cost}
    other = tot - sum(cl.values())
    cycles = sum(cl[c] * cost[c] for c in cost) + other * 3.43
    out["valu_classes_per_64_samples"] = {**{c.lower(): round(v, 2) for c, v in cl.items()}, "other": round(other, 2), "total": round(tot, 2),
                                          "salu": round(per.get("SQ_INSTS_SALU", 0.0), 2), "smem": round(per.get("SQ_INSTS_SMEM", 0.0), 2),
                                          "lds": round(per.get("SQ_INSTS_LDS", 0.0), 2), "vmem_rd": round(per.get("SQ_INSTS_VMEM_RD", 0.0), 2),
                                          "branch": round(per.get("SQ_INSTS_BRANCH", 0.0), 2)}
    out["issue_cycles_per_inst"] = round(cycles / tot, 3)
    out["f64_inst_frac"] = round((cl["ADD_F64"] + cl["MUL_F64"] + cl["FMA_F64"] + cl["TRANS_F64"]) / tot, 4)
main = None
for name, cs in counters.items():
    k = {c: sum(v) / len(v) for c, v in cs.items()}
    k.update(meta[name])
    out["kernels"][name] = k
    if (kname and kname in name and "<true" not in name) or ("<false>" in name and not kname) or main is None:
        main = k
if main and "FETCH_SIZE" in main and "WRITE_SIZE" in main:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in units of 1024 B
    out["l2_miss_bytes_per_launch_raw"] = (main["FETCH_SIZE"] + main["WRITE_SIZE"]) * 1024.0
    # correction factor for FETCH_SIZE: measured by the calibration kernels of this very run when present
    # (both of this kernel's access widths, 16 B and 8 B per lane, read exactly 1/2 on gfx950), else argv/1.0
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    if calib and len(sys.argv) <= 2:
        ratios = [float(1 << 30) / x for v in calib.values() for x in v if x]
        scale = sum(ratios) / len(ratios)
    out["fetch_scale"] = scale
    # bytes that left the L2s towards the fabric: one 128-B line per miss, tallied at 64 B by FETCH_SIZE (streams AND divergent
    # gathers: profiles/r03_gather_calibration.json); Infinity-Cache hits are counted, so this bounds HBM bytes from above
    out["l2_miss_bytes_per_launch"] = (main["FETCH_SIZE"] * scale + main["WRITE_SIZE"]) * 1024.0
    out["hbm_bytes_per_launch"] = None  # no DRAM-side counter on gfx950
if main and "SQ_ACTIVE_INST_VALU" in main and "SQ_BUSY_CYCLES" in main and main["SQ_BUSY_CYCLES"]:
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; SQ_BUSY_CYCLES is summed over the 32 shader
    # engines (MI355X_MICROARCH.md): busy fraction of the 1024 SIMDs' cycles spent issuing VALU instructions
    cycles = main["SQ_BUSY_CYCLES"] / 32.0
    out["valu_busy_frac"] = main["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * 1024.0)
    if "SQ_INSTS_VALU" in main and "SQ_WAVES" in main:
        out["valu_insts_per_launch"] = main["SQ_INSTS_VALU"]
if main and "SQ_THREAD_CYCLES_VALU" in main and main.get("SQ_ACTIVE_INST_VALU@2"):
    # lanes active per issued VALU instruction: thread-cycles / (64 lanes x instruction cycles), both from one pass
    out["lanes_active_frac"] = main["SQ_THREAD_CYCLES_VALU"] / (64.0 * main["SQ_ACTIVE_INST_VALU@2"])
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench_lines"}, indent=1)[:3000])
