#!/usr/bin/env python3
"""Lane utilisation of the BVH kernel's sections (experiment build -DFLUX_DEBUG_TRIPS via FLUX_HIP_LIB): wave-level
executions and active lanes of the shading section, the node step and the triangle test.
usage: FLUX_HIP_LIB=flux_amd/variants/libflux_hip_trips.so python scripts/bvh_lanes.py [NXxNZ] [root]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux_amd
from flux_amd.procedural import heightfield_scene
nx, nz = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1000x500").split("x")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sd = heightfield_scene(nx, nz)
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
r.enable_stats(True); r.stats(reset=True)
r.render_frame()
raw = r.stats_raw()
seg = raw[1]
print(f"kernel {r.last_kernel_ms():.2f} ms (instrumented); segments {seg}, nodes/segment {raw[8] / seg:.2f}, tris/segment {raw[9] / seg:.2f}")
for name, k in (("shade", 10), ("node step", 12), ("triangle test", 14)):
    trips, lanes = raw[k], raw[k + 1]
    print(f"  {name:14s} wave executions per 64 segments {trips / (seg / 64.0):8.2f}   lanes active {lanes / max(trips, 1) / 64.0:6.3f}")
