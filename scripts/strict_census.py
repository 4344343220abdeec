#!/usr/bin/env python3
"""STRICT's per-lane candidate loop, counted (experiment build with -DFLUX_DEBUG_SCENSUS, selected via FLUX_HIP_LIB):
usage: FLUX_HIP_LIB=flux_amd/variants/libflux_hip_scensus.so python scripts/strict_census.py [scene] [root]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux_amd
scene = sys.argv[1] if len(sys.argv) > 1 else "demo2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sd = flux_amd.load_scene(f"scenes/{scene}.yml")
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
r.set_math(flux_amd.MATH_STRICT)
r.set_kernel(2)
r.enable_stats(True); r.stats(reset=True)
r.render_frame()
raw = r.stats_raw()
scans, scan_lanes, trips, trip_lanes, quads, quad_lanes = raw[10:16]
print(f"STRICT refill, {scene} n={n}: scans {scans} at {scan_lanes / max(scans, 1):.1f} lanes; candidate trips per scan {trips / max(scans, 1):.2f} at "
      f"{trip_lanes / max(trips, 1):.1f} lanes (candidates per scanning lane {trip_lanes / max(scan_lanes, 1):.2f}); quadratics per scan "
      f"{quads / max(scans, 1):.2f} at {quad_lanes / max(quads, 1):.1f} lanes")
print(f"  lane-slots in trips {trips * 64} of which used {trip_lanes} = {trip_lanes / max(trips * 64, 1):.3f}; if compacted: "
      f"{trip_lanes / 64 / max(scans, 1):.2f} full trips per scan instead of {trips / max(scans, 1):.2f}")
