#!/usr/bin/env python3
"""Static ISA summary of the render kernels (no GPU needed): registers, spills, occupancy and the
instruction mix of each kernel's body, from `hipcc -S` of csrc/render.hip with the build's flags.
usage: scripts/isa_mix.py [out.txt]   (default profiles/isa_mix.txt)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux_amd"))
import importlib.util
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "flux_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_mix.txt")
flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "render.s")
    cmd = [b._hipcc()] + flags + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", asm,
                                  os.path.join(b.CSRC, "render.hip")]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=b.CSRC)
    if p.returncode:
        sys.exit(p.stderr)
    remarks = p.stderr
    text = open(asm).read()

res = collections.OrderedDict()
cur = None
for line in remarks.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = res.setdefault(m.group(1), {})
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))

demangle = subprocess.run(["c++filt"] + list(res), capture_output=True, text=True).stdout.split("\n")
names = dict(zip(res, demangle))

lines = ["# static ISA summary of csrc/render.hip (flags: " + " ".join(flags) + ")", ""]
for mangled, r in res.items():
    body = re.search(r"^%s:[^\n]*\n(.*?)s_endpgm" % re.escape(mangled), text, re.S | re.M)
    if not body or "render_" not in mangled:
        continue
    ops = collections.Counter()
    for ln in body.group(1).splitlines():
        m = re.match(r"\s+([a-z][a-z0-9_]+)", ln)
        if m:
            ops[m.group(1)] += 1
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    salu = sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith(("s_load", "s_waitcnt", "s_nop")))
    smem = sum(v for k, v in ops.items() if k.startswith("s_load"))
    vmem = sum(v for k, v in ops.items() if k.startswith(("global_", "scratch_", "buffer_", "flat_")))
    lds = sum(v for k, v in ops.items() if k.startswith("ds_"))
    f64 = sum(v for k, v in ops.items() if k.startswith("v_") and "f64" in k)
    lines.append(names.get(mangled, mangled).replace("(flux::RenderParams)", ""))
    lines.append("  VGPRs %d  SGPRs %d  scratch %d B/lane  VGPR spills %d  occupancy %d waves/SIMD" % (
        r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", 0), r.get("VGPRs Spill", 0),
        r.get("Occupancy", -1)))
    lines.append("  static instructions: VALU %d (f64 %d)  SALU %d  SMEM %d  VMEM %d  LDS %d" % (valu, f64, salu, smem, vmem, lds))
    top = ", ".join("%s %d" % kv for kv in ops.most_common(12))
    lines.append("  top: " + top)
    lines.append("")
open(out_path, "w").write("\n".join(lines))
print("\n".join(lines[:40]))
