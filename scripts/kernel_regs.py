#!/usr/bin/env python3
"""Compile ONE render kernel (no GPU needed) and print its registers, spills and occupancy -- the quick loop of a register
diet: ~15 s instead of a full build.  usage: scripts/kernel_regs.py <kernel> [--strict] [-Dflag ...] [--asm out.s] [--csrc dir]
   kernel: bvh4 | bvh | split | refill | static   (FAST arithmetic, the product instantiation: no statistics)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import importlib.util
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "flux_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

INST = {"bvh4": "render_bvh4_kernel<false, true, true>", "bvh4l": "render_bvh4_kernel<false, true, false>", "bvh4g": "render_bvh4_kernel<false, false, false>", "bvh": "render_bvh_kernel<false>", "split": "render_split_kernel<false, true, true>", "split32": "render_split_kernel<false, true, false>", "split64": "render_split_kernel<false, false, false>",
        "refill": "render_refill_kernel<false, false>", "static": "render_static_kernel<false, false>",
        "bvh4s": "render_bvh4_kernel<true, true, true>", "refill_tris": "render_refill_kernel<false, true>", "static_tris": "render_static_kernel<false, true>"}
args = sys.argv[1:]
asm_out = None
if "--asm" in args:
    k = args.index("--asm")
    asm_out = args[k + 1]
    del args[k:k + 2]
csrc = b.CSRC
if "--csrc" in args:   # another checkout's flux_amd/csrc (A/B against an older kernel)
    k = args.index("--csrc")
    csrc = os.path.abspath(args[k + 1])
    del args[k:k + 2]
strict = "--strict" in args   # the STRICT arithmetic's instantiation of the same kernel (namespace flux::strict, no contraction)
if strict:
    args.remove("--strict")
kernel, extra = args[0], args[1:]
render = open(os.path.join(csrc, "render.hip")).read()
head = render[:render.index("// The loop itself lives in render_body.inc")]  # includes + tunables
src = head + (f'''
#define FLUX_FAST 1
#define FLUX_WPE FLUX_WAVES_PER_EU_FAST
#define FLUX_WPE_WIDE FLUX_WAVES_PER_EU_FAST_WIDE
#define FLUX_EXP_NO_LAUNCH 1
#pragma clang fp contract(fast)
namespace flux {{
namespace fast {{
#include "render_body.inc"
template __global__ void {INST[kernel]}(const RenderParams);
}}
}}
''' if not strict else f'''
#define FLUX_FAST 0
#define FLUX_WPE FLUX_WAVES_PER_EU
#define FLUX_WPE_WIDE FLUX_WAVES_PER_EU
#define FLUX_EXP_NO_LAUNCH 1
#pragma clang fp contract(off)
namespace flux {{
namespace strict {{
#include "render_body.inc"
template __global__ void {INST[kernel]}(const RenderParams);
}}
}}
''')
flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(csrc, "_kernel_regs_tmp.hip")
    open(path, "w").write(src)
    asm = asm_out or os.path.join(td, "k.s")
    try:
        p = subprocess.run([b._hipcc()] + flags + extra + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", asm, path],
                           capture_output=True, text=True, cwd=csrc)
    finally:
        os.unlink(path)
    if p.returncode:
        sys.exit(p.stderr[-4000:])
    keep = ("Function Name", "VGPRs:", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill", "TotalSGPRs")
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if m and any(k in m.group(1) for k in keep):
            print(m.group(1).strip())
    text = open(asm).read()
    body = text[text.index("s_load") if "s_load" in text else 0:]
    n_valu = len(re.findall(r"^\s+v_", body, flags=re.M))
    print("static VALU instructions:", n_valu, " v_mov:", len(re.findall(r"^\s+v_mov_b", body, flags=re.M)),
          " scratch_:", len(re.findall(r"^\s+scratch_", body, flags=re.M)))
