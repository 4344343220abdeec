"""STRICT frames of two builds of the library, bit for bit (GPU box): the shipped build (STRICT's sphere candidates from the
conservative f32 filter, FLUX_STRICT_FILTER=1) against a variant with the full scan (-DFLUX_STRICT_FILTER=0
-DFLUX_STRICT_BOX_HWMINMAX=0: round 4's scan).  Each build renders in its own process (one library per process).
usage: scripts/strict_filter_check.py <variant.so>      -> prints one line per scene, exits 1 on any difference"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, copy
sys.path.insert(0, %r)
import numpy as np
import flux_amd
sys.path.insert(0, os.path.join(%r, "tests"))
from test_gpu_fuzz import random_scene
out = sys.argv[1]
frames = {}
def render(tag, sd, n, D, seed):
    with flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, D, 50), seed=seed) as r:
        r.set_math(flux_amd.MATH_STRICT)
        for v in (flux_amd.KERNEL_STATIC, flux_amd.KERNEL_REFILL):
            r.set_kernel(v)
            r.enable_stats(True); r.stats(reset=True)
            frames[f"{tag}/k{v}"] = r.render_frame()
            st = r.stats()
            frames[f"{tag}/k{v}/stats"] = np.array([st[k] for k in sorted(st)], dtype=np.int64)
for name in ("demo1", "demo2"):
    sd = flux_amd.load_scene(os.path.join(%r, "scenes", name + ".yml"))
    sd = copy.deepcopy(sd)
    sd.output_settings.image_width, sd.output_settings.image_height = 160, 120
    sd.output_settings.pixel_size *= 5
    render(name, sd, 8, 5, 1)
demo1 = flux_amd.load_scene(os.path.join(%r, "scenes", "demo1.yml"))
rng = np.random.default_rng(4242)
for case in range(int(os.environ.get("FLUX_CHECK_SCENES", "200"))):
    sd = random_scene(flux_amd, demo1, rng, unit_planes=case %% 2 == 1)
    render(f"fuzz{case}", sd, int(rng.choice([1, 2, 3, 8])), int(rng.choice([1, 3, 5, 9])), int(rng.integers(1, 1 << 30)))
np.savez(out, **frames)
''' % (ROOT, ROOT, ROOT, ROOT)


def run(lib, out):
    env = dict(os.environ)
    if lib:
        env["FLUX_HIP_LIB"] = os.path.abspath(lib)
    else:
        env.pop("FLUX_HIP_LIB", None)
    subprocess.run([sys.executable, "-c", CHILD, out], check=True, env=env, cwd=ROOT)
    return np.load(out)


if __name__ == "__main__":
    a = run(None, "/tmp/strict_a.npz")
    b = run(sys.argv[1], "/tmp/strict_b.npz")
    bad = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        bad += not same
        if not same or not k.startswith("fuzz"):
            print(f"{k}: {'identical' if same else 'DIFFERENT'}")
    print(f"{len(a.files)} arrays (frames + path statistics, STRICT, static and refill kernels): {bad} differ")
    sys.exit(1 if bad else 0)
