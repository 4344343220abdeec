#!/usr/bin/env python3
"""Per-basic-block map of a kernel's assembly (scripts/kernel_asm.py <kernel> --asm out.s): instructions, VALU / VMEM / LDS
counts, distinct VGPRs referenced and the highest one -- where a kernel's register peak sits.  usage: asm_blocks.py out.s [min_insts]"""
import re
import sys

t = open(sys.argv[1]).read().splitlines()
min_insts = int(sys.argv[2]) if len(sys.argv) > 2 else 12
start = [i for i, l in enumerate(t) if re.match(r"^_ZN.*:", l)][0]
blocks, cur = [], ("entry", [], "")
for l in t[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [], m.group(2).strip())
    else:
        cur[1].append(l)
blocks.append(cur)


def vregs(l):
    s = set()
    for m in re.finditer(r"\bv(\d+)\b", l):
        s.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", l):
        s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return s


for name, ls, note in blocks:
    ins = [l for l in ls if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(";")]
    used = set()
    for l in ins:
        used |= vregs(l)
    nv = sum(1 for l in ins if l.strip().startswith("v_"))
    vm = sum(1 for l in ins if re.match(r"\s+(global_|buffer_|flat_|scratch_)", l))
    ds = sum(1 for l in ins if l.strip().startswith("ds_"))
    if len(ins) >= min_insts:
        print(f"{name:12s} insts {len(ins):4d} valu {nv:4d} vmem {vm:3d} lds {ds:3d} distinct vregs {len(used):3d} max v{max(used) if used else -1:<4d} {note[:60]}")
