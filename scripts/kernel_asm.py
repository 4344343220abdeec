#!/usr/bin/env python3
"""One render kernel's registers, spills, occupancy and (optionally) its assembly, without a GPU: `hipcc -S --cuda-device-only` of
csrc/render.hip with the build's flags (~15 s), then the named instantiation cut out of the listing.
usage: scripts/kernel_asm.py <kernel> [--strict] [-Dflag ...] [--asm out.s] [--csrc dir]
   kernel: split | split32 | split64 | refill | static | bvh4 | bvh4l | bvh4g | bvh | ... (INST below), or any substring of a demangled name
scripts/asm_blocks.py out.s maps the basic blocks of the result."""
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

INST = {"bvh4": "render_bvh4_kernel<false, true, true>", "bvh4l": "render_bvh4_kernel<false, true, false>",
        "bvh4g": "render_bvh4_kernel<false, false, false>", "bvh": "render_bvh_kernel<false>",
        "split": "render_split_kernel<false, true, true>", "split32": "render_split_kernel<false, true, false>",
        "split64": "render_split_kernel<false, false, false>", "refill": "render_refill_kernel<false, false>",
        "static": "render_static_kernel<false, false>", "bvh4s": "render_bvh4_kernel<true, true, true>",
        "splits": "render_split_kernel<true, true, true>",
        "refill_tris": "render_refill_kernel<false, true>", "static_tris": "render_static_kernel<false, true>"}
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
ns = "strict" if "--strict" in args else "fast"
if "--strict" in args:
    args.remove("--strict")
want = INST.get(args[0], args[0])
extra = args[1:]
flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "render.s")
    p = subprocess.run([b._hipcc()] + flags + extra + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", asm,
                                                       os.path.join(csrc, "render.hip")], capture_output=True, text=True, cwd=csrc)
    if p.returncode:
        sys.exit(p.stderr[-4000:])
    text = open(asm).read()
# mangled names of the kernels in the listing, demangled
names = re.findall(r"^\s*\.globl\s+(_Z\S+)", text, flags=re.M)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
pick = [m for m, d in zip(names, dem) if f"flux::{ns}::" in d and want in d]
if len(pick) != 1:
    sys.exit(f"{len(pick)} kernels match {want!r} in flux::{ns}:\n" + "\n".join(d for d in dem if "render_" in d))
sym = pick[0]
cur = None
keep = ("VGPRs:", "AGPRs", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill", "TotalSGPRs", "LDS Size")
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        if cur == sym:
            print("Function Name:", dem[names.index(sym)])
        continue
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if m and cur == sym and any(k in m.group(1) for k in keep):
        print(m.group(1).strip())
start = text.index(f"\n{sym}:")
end = text.index(".Lfunc_end", start)
end = text.index("\n", end)
body = text[start + 1:end + 1]
if asm_out:
    open(asm_out, "w").write(body)
cnt = lambda pat: len(re.findall(pat, body, flags=re.M))  # noqa: E731
P = {k: cnt(v) for k, v in dict(valu=r'^\s+v_', mov=r'^\s+v_mov_b', f64=r'^\s+v_\w+_f64', lane=r'^\s+v_(read|write)lane', salu=r'^\s+s_(?!load|waitcnt|barrier|nop|endpgm|branch|cbranch|buffer)', smem=r'^\s+s_(load|buffer_load)', br=r'^\s+s_c?branch', vmem=r'^\s+(global|buffer|flat)_', lds=r'^\s+ds_', scratch=r'^\s+scratch_').items()}
print("static instructions: VALU {valu} (v_mov {mov}, f64 {f64}, v_readlane/writelane {lane}) SALU {salu} SMEM {smem} branches {br} "
      "VMEM {vmem} LDS {lds} scratch {scratch}".format(**P))
