#!/usr/bin/env python3
"""A/B of library builds on the GPU box: kernel time (min and median of `reps` frames after a warm-up) and the SHA-1 of the frame,
each library in a fresh child process, round-robin over `rounds` so that clock / thermal drift hits all alike.
usage: scripts/ab_frames.py <scene> <root> <kernel variant> <rounds> name [name ...]      name = default | a file under flux_amd/variants/
       (libflux_hip_<name>.so or <name>.so)"""
import hashlib
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib
sys.path.insert(0, %r)
import flux_amd
scene, n, variant, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if scene.startswith("hf:"):
    from flux_amd.procedural import heightfield_scene
    nx, nz = [int(x) for x in scene[3:].split("x")]
    sd = heightfield_scene(nx, nz)
else:
    sd = flux_amd.load_scene(os.path.join(%r, "scenes", scene + ".yml"))
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
r.set_kernel(variant)
img = r.render_frame()
ms = []
for _ in range(reps):
    img = r.render_frame(); ms.append(r.last_kernel_ms())
print("RESULT", hashlib.sha1(img.tobytes()).hexdigest()[:12], " ".join("%%.3f" %% m for m in ms))
''' % (ROOT, ROOT)


def lib_path(name):
    if name == "default":
        return None
    for cand in (os.path.join(ROOT, "flux_amd", "variants", f"libflux_hip_{name}.so"), os.path.join(ROOT, "flux_amd", "variants", f"{name}.so")):
        if os.path.exists(cand):
            return cand
    sys.exit(f"no such variant: {name}")


scene, root, variant, rounds = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
names = sys.argv[5:]
times = {n: [] for n in names}
sha = {}
for rnd in range(rounds):
    for name in names:
        env = dict(os.environ)
        env.pop("FLUX_HIP_LIB", None)
        if lib_path(name):
            env["FLUX_HIP_LIB"] = lib_path(name)
        p = subprocess.run([sys.executable, "-c", CHILD, scene, root, variant, "3"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(f"{name}: FAILED {p.stderr[-300:]}", flush=True)
            continue
        tok = line[0].split()
        sha[name] = tok[1]
        times[name] += [float(x) for x in tok[2:]]
base = names[0]
for name in names:
    if not times[name]:
        continue
    t = times[name]
    rel = (statistics.median(t) / statistics.median(times[base]) - 1.0) * 100.0 if times[base] else 0.0
    print(f"{name:28s} min {min(t):8.3f} ms  median {statistics.median(t):8.3f} ms  ({rel:+.2f} % vs {base})  frame {sha[name]}"
          f"{'' if sha[name] == sha[base] else '  DIFFERENT FRAME'}", flush=True)
