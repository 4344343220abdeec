#!/usr/bin/env python3
"""A/B of whole TREES (package + library) on the GPU box, for builds whose ABI differs: each tree's flux_amd is imported in a fresh
child, `rounds` times round-robin.  usage: scripts/ab_trees.py <scene> <root> <kernel variant> <rounds> <tree dir> [<tree dir> ...]
(a tree dir holds flux_amd/ with its libflux_hip.so and scenes/; `.` = this checkout)"""
import os, statistics, subprocess, sys
CHILD = r'''
import sys, os, hashlib
tree = os.path.abspath(sys.argv[1]); sys.path.insert(0, tree)
import flux_amd
scene, n, variant = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
if scene.startswith("hf:"):
    from flux_amd.procedural import heightfield_scene
    nx, nz = [int(x) for x in scene[3:].split("x")]
    sd = heightfield_scene(nx, nz)
else:
    sd = flux_amd.load_scene(os.path.join(tree, "scenes", scene + ".yml"))
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
r.set_kernel(variant)
img = r.render_frame(); ms = []
for _ in range(3):
    img = r.render_frame(); ms.append(r.last_kernel_ms())
print("RESULT", hashlib.sha1(img.tobytes()).hexdigest()[:12], " ".join("%.3f" % m for m in ms))
'''
scene, root, variant, rounds = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
trees = sys.argv[5:]
times = {t: [] for t in trees}; sha = {}
for _ in range(rounds):
    for t in trees:
        env = dict(os.environ); env.pop("FLUX_HIP_LIB", None)
        p = subprocess.run([sys.executable, "-c", CHILD, t, scene, root, variant], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(t, "FAILED", p.stderr[-300:]); continue
        tok = line[0].split(); sha[t] = tok[1]; times[t] += [float(x) for x in tok[2:]]
base = trees[0]
for t in trees:
    if times[t]:
        m = statistics.median(times[t])
        print(f"{t:40s} min {min(times[t]):9.3f} ms  median {m:9.3f} ms  ({(m / statistics.median(times[base]) - 1) * 100:+.2f} % vs {base})  frame {sha[t]}"
              f"{'' if sha[t] == sha[base] else '  DIFFERENT FRAME'}", flush=True)
