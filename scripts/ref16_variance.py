#!/usr/bin/env python3
"""Why is demo.png less noisy than a 16384-spp render of the restated estimator?  (DESIGN.md "Oracle")

Measures, per image region, the per-pixel variance of (a) the reference image about the 16-seed mean of the default
build at 16384 spp, and (b) single renders of candidate configurations about the same mean:
    default build at sample_root 128 (the null case), at sample_root 256 (4x the samples), and an EXPERIMENT build
    whose hemisphere stream uses the correlated multi-jittered structure (FLUX_EXP_HEMI_CMJ).
Each candidate runs in a child process (FLUX_HIP_LIB selects the library at import).
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "gpurun_out")

CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
import flux_amd
root, seeds, out = int(sys.argv[1]), [int(s) for s in sys.argv[2].split(",")], sys.argv[3]
sd = flux_amd.load_scene(os.path.join(%r, "scenes", "demo2.yml"))
frames = []
for seed in seeds:
    with flux_amd.Renderer(sd, flux_amd.JobConfiguration(root, 5, 50), seed=seed) as r:
        frames.append(r.render_frame())
np.save(out, np.asarray(frames))
''' % (ROOT, ROOT)


def render(lib, root, seeds, tag):
    out = os.path.join(OUT, f"ref16_var_{tag}.npy")
    env = dict(os.environ)
    if lib:
        env["FLUX_HIP_LIB"] = lib
    subprocess.run([sys.executable, "-c", CHILD, str(root), ",".join(map(str, seeds)), out], env=env, check=True)
    x = np.load(out)
    os.remove(out)
    return x


def main():
    import flux_amd
    import ref16
    from scipy.ndimage import binary_erosion
    os.makedirs(OUT, exist_ok=True)
    M = 16
    base = render(None, 128, list(range(2, M + 2)), "base")
    mean, var = ref16.seed_moments(base)
    ref = ref16.load_ref16()
    sd = flux_amd.load_scene(os.path.join(ROOT, "scenes", "demo2.yml"))
    omap = ref16.object_map(sd)
    cands = {"reference demo.png": ref[None]}
    cands["default root128 (null)"] = render(None, 128, [101, 102], "null")
    cands["default root256 (65536 spp)"] = render(None, 256, [101, 102], "r256")
    cands["default root64 (4096 spp)"] = render(None, 64, [101, 102], "r64")
    var_lib = os.path.join(ROOT, "flux_amd", "variants", "libflux_hip_hemicmj.so")
    if os.path.exists(var_lib):
        cands["hemi-CMJ experiment root128"] = render(var_lib, 128, [101, 102], "hcmj")
    regions = [(int(k), binary_erosion(omap == k, iterations=3)) for k in np.unique(omap)]
    print("region sizes:", {k: int(m.sum()) for k, m in regions})
    print("v_x / v_default16384 per region (mean over channels), v_x = E[(x - mean16)^2] - v/16:")
    for name, xs in cands.items():
        row = []
        for k, m in regions:
            if m.sum() < 100:
                continue
            v = var[m].mean()
            e = np.mean([((x - mean)[m] ** 2).mean() for x in xs])
            row.append((e - v / M) / v)
        gm = np.mean([x.mean() for x in xs])
        print(f"{name:32s} " + " ".join(f"{r:6.3f}" for r in row) + f"   image mean {gm:.6f} (mean16 {mean.mean():.6f})")


if __name__ == "__main__":
    main()
