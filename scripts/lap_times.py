#!/usr/bin/env python3
"""Where the split kernel's cycles go (experiment build -DFLUX_DEBUG_CLOCK via FLUX_HIP_LIB): lap-timer totals of
lane 0 of every wave, as fractions.  usage: FLUX_HIP_LIB=... python scripts/lap_times.py [scene] [root]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux_amd
scene = sys.argv[1] if len(sys.argv) > 1 else "demo2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sd = flux_amd.load_scene(f"scenes/{scene}.yml")
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
r.set_kernel(3)
r.enable_stats(True); r.stats(reset=True)
r.render_frame()
raw = r.stats_raw()
names = ["pop", "phase A", "B plane+setup", "B filter", "B candidates", "B shade+rest"]
tot = sum(raw[10:16])
print(f"kernel {r.last_kernel_ms():.2f} ms (instrumented)")
for k, nm in enumerate(names):
    print(f"  {nm:16s} {100.0 * raw[10 + k] / tot:5.1f} %   {raw[10 + k] / (raw[0] / 64.0):8.1f} wave-cycles per 64 samples")
