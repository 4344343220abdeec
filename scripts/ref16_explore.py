#!/usr/bin/env python3
"""Exploration behind tests/test_gpu_ref16.py: render demo2 at 16384 spp with M seeds, measure the estimator's
per-pixel variance and print every statistic of the 16-bit reference comparison, for the reference and for a
held-out seed in its place.  Writes gpurun_out/ref16_moments.npz for offline analysis."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flux_amd  # noqa: E402
import ref16  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
root = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sd = flux_amd.load_scene(os.path.join(ROOT, "scenes", "demo2.yml"))
cfg = flux_amd.JobConfiguration(root, 5, 50)
frames = []
t0 = time.time()
for seed in range(1, M + 2):
    with flux_amd.Renderer(sd, cfg, seed=seed) as r:
        frames.append(r.render_frame())
    print(f"seed {seed} done {time.time() - t0:.1f}s", flush=True)
hold = frames[0]
mean, var = ref16.seed_moments(frames[1:])
ref = ref16.load_ref16()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "ref16_moments.npz"), mean=mean, var=var.astype(np.float32),
                    hold=hold)
omap = ref16.object_map(sd)
names = {-1: "none", 0: "sky", 1: "light", 12: "floor"}
for label, x in (("REF", ref), ("HOLD", hold)):
    print("=====", label)
    diff, se, n = ref16.aggregate_stats(x, mean, var, M)
    print("whole image: diff", diff, "se", se, "z", diff / se, "all-channel diff", diff.mean())
    for k in sorted(np.unique(omap)):
        diff, se, n = ref16.aggregate_stats(x, mean, var, M, omap == k)
        print(f"region {names.get(int(k), 'sphere%d' % k):9s} n={n:6d} diff {diff} se {se} z {diff / se}")
    z, noisy = ref16.zscores(x, mean, var, M)
    zz = z[noisy]
    print("noisy frac", noisy.mean(), "z mean", zz.mean(), "std", zz.std(), "|z|>2", (np.abs(zz) > 2).mean(), "|z|>3",
          (np.abs(zz) > 3).mean(), "|z|>5", (np.abs(zz) > 5).mean(), "max", np.abs(zz).max())
    quiet = ~noisy
    if quiet.any():
        print("quiet pixels max |d|", np.abs((x - mean)[quiet]).max(), "in quanta", np.abs((x - mean)[quiet]).max() * 65535.99)
    d8 = (x - mean).reshape(75, 8, 100, 8, 3).mean(axis=(1, 3))
    v8 = (var * (1 + 1 / M) + ref16.QUANT_VAR).reshape(75, 8, 100, 8, 3).sum(axis=(1, 3)) / 64 ** 2
    z8 = d8 / np.sqrt(v8)
    print("8x8: mean|d|", np.abs(d8).mean(), "max|d|", np.abs(d8).max(), "p99", np.percentile(np.abs(d8), 99), "z8 std",
          z8.std(), "z8 max", np.abs(z8).max())
    print("per-pixel |d|: mean", np.abs(x - mean).mean(), "p99", np.percentile(np.abs(x - mean), 99), "max",
          np.abs(x - mean).max())
