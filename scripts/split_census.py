#!/usr/bin/env python3
"""Lane utilisation of the split kernel's sections (experiment builds -DFLUX_DEBUG_CENSUS=0|3|6|9 under flux_amd/variants/census_*.so).
Sections: 0 phase-A pass, 1 phase-A sphere test, 2 phase-A shading entry, 3 phase-B pass (scan + shade), 4 candidate-loop trip,
5 wave-uniform invert-sphere test, 6/7/8 phase-B lobe code / Matte part / Glossy part, 9/10/11 the same three in phase A.
usage: python scripts/split_census.py [scene] [root]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "demo2"
n = sys.argv[2] if len(sys.argv) > 2 else "32"
NAMES = ["A pass", "A sphere test", "A shading entry", "B pass", "B candidate trip", "B invert-sphere test", "B lobe code", "B lobe: Matte part",
         "B lobe: Glossy part", "A lobe code", "A lobe: Matte part", "A lobe: Glossy part"]
code = r'''
import sys, os
sys.path.insert(0, %r)
import flux_amd
sd = flux_amd.load_scene(os.path.join(%r, "scenes", %r + ".yml"))
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(%s, 5, 50), seed=1)
r.set_kernel(3); r.enable_stats(True); r.stats(reset=True); r.render_frame()
print("RAW", r.stats_raw())
''' % (ROOT, ROOT, scene, n)
for lib in sorted(glob.glob(os.path.join(ROOT, "flux_amd", "variants", "census_*.so"))):
    base = int(os.path.basename(lib)[7:-3])
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FLUX_HIP_LIB=lib), capture_output=True, text=True)
    raw = eval([l for l in p.stdout.splitlines() if l.startswith("RAW")][0][4:])
    groups = raw[0] / 64.0
    for k in range(3):
        ex, lanes = raw[10 + 2 * k], raw[11 + 2 * k]
        if base + k < len(NAMES):
            print(f"section {base + k:2d} {NAMES[base + k]:24s} executions per 64 samples {ex / groups:7.3f}   lanes active {lanes / max(ex, 1) / 64.0:6.3f}")
