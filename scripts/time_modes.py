"""Kernel time of MATH_FAST vs MATH_STRICT (development aid). usage: time_modes.py <scene> <root> [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flux_amd
scene = sys.argv[1] if len(sys.argv) > 1 else "demo2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if scene.startswith("hf:"):
    from flux_amd.procedural import heightfield_scene
    nx, nz = [int(x) for x in scene[3:].split("x")]
    sd = heightfield_scene(nx, nz)
else:
    sd = flux_amd.load_scene(f"scenes/{scene}.yml")
W, H = sd.output_settings.image_width, sd.output_settings.image_height
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
imgs = {}
for name, mode in (("strict", flux_amd.MATH_STRICT), ("fast", flux_amd.MATH_FAST)):
    r.set_math(mode)
    for v in (1, 2):
        r.set_kernel(v)
        best = 1e30
        for _ in range(reps):
            img = r.render_frame(); best = min(best, r.last_kernel_ms())
        imgs[(name, v)] = img
        print(f"{scene} n={n} {name:6s} variant {v}: {best:9.2f} ms  {W*H*n*n/best/1e3:9.1f} Msamples/s", flush=True)
d = np.abs(imgs[("fast", 2)] - imgs[("strict", 2)])
print(f"fast vs strict (refill): max |d| = {d.max():.3e}, 99.9th pct = {np.percentile(d, 99.9):.3e}, mean = {d.mean():.3e}")
