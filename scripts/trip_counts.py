#!/usr/bin/env python3
"""Loop trip counts of the render kernels (experiment build with -DFLUX_DEBUG_TRIPS, selected via FLUX_HIP_LIB).
usage: FLUX_HIP_LIB=flux_amd/variants/libflux_hip_trips.so python scripts/trip_counts.py [scene] [root]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux_amd
scene = sys.argv[1] if len(sys.argv) > 1 else "demo2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sd = flux_amd.load_scene(f"scenes/{scene}.yml")
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
for v, name in ((2, "refill"), (3, "split")):
    r.set_kernel(v)
    r.enable_stats(True); r.stats(reset=True)
    r.render_frame()
    raw = r.stats_raw()
    per = lambda x: x / (raw[0] / 64.0)
    print(f"{name}: per 64 samples: lane-segments {per(raw[1]):.1f}  loop iterations {per(raw[10]):.2f}  candidate-loop trips "
          f"{per(raw[11]):.2f}  phase-A passes {per(raw[12]):.2f}  phase-A sphere tests {per(raw[13]):.2f}  raw {raw[10:14]}")
