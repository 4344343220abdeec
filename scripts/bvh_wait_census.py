"""What the idle lanes of the mesh kernel's node steps wait for (experiment build -DFLUX_DEBUG_TRIPS -DFLUX_DEBUG_WAIT via FLUX_HIP_LIB):
per node step, lanes traversing / holding a leaf until the vote / marked for shading / finished but not yet marked or out of samples.
usage: FLUX_HIP_LIB=flux_amd/variants/t_trips_wait.so python scripts/bvh_wait_census.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import flux_amd
from flux_amd.procedural import heightfield_scene
sd = heightfield_scene(1000, 500)
for n in (16, 32):
    r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
    r.enable_stats(True); r.stats(reset=True)
    r.render_frame()
    raw = r.stats_raw()
    seg = raw[1]
    trips, act = raw[12], raw[13]
    print(f"n={n}: segments {seg}; node steps per 64 segments {trips/(seg/64):.2f}; per step: traversing {act/trips:.1f} lanes, holding a leaf {raw[14]/trips:.1f}, "
          f"waiting to be shaded {raw[15]/trips:.1f}, finished, not yet marked (or out of samples) {64 - (act+raw[14]+raw[15])/trips:.1f}; shade steps per 64 segments {raw[10]/(seg/64):.2f} at {raw[11]/max(raw[10],1):.1f} lanes")
    r.close()
