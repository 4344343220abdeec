#!/usr/bin/env python3
"""One-off (GPU box): the WHOLE headline frame -- demo2.yml 800x600 @16384 spp, every row -- against the CPU checker on all host threads
(~2 minutes of oracle), FAST and STRICT, product build (statistics off): max / 99.9th percentile of |gpu - oracle| per channel and the
whole-image means.  The test suite compares every 25th row (tests/test_gpu_whole_frames.py); this is the same comparison without the stride.
usage: scripts/full_frame_check.py [sample_root] [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flux_amd as flux
from oracle import oracle as oracle_mod
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 16)
sd = flux.load_scene(os.path.join(ROOT, "scenes", "demo2.yml"))
cfg = flux.JobConfiguration(n, 5, 50)
t0 = time.time()
o = oracle_mod.Oracle(sd, cfg, seed=1)
want = np.asarray(o.render_frame(threads=threads))
o.close()
print(f"oracle: {want.shape} in {time.time() - t0:.1f} s on {threads} threads, mean {want.mean(axis=(0, 1))}", flush=True)
with flux.Renderer(sd, cfg, seed=1) as r:
    for name, math in (("FAST", flux.MATH_FAST), ("STRICT", flux.MATH_STRICT)):
        r.set_math(math)
        got = np.asarray(r.render_frame())
        d = np.abs(got - want)
        print(f"{name}: kernel {r.last_kernel_ms():.1f} ms  max |gpu - oracle| {d.max():.3e}  99.9th pct {np.quantile(d, 0.999):.3e}  "
              f"mean diff {np.abs(got.mean(axis=(0, 1)) - want.mean(axis=(0, 1))).max():.3e}  nan {int(np.isnan(got).sum())}/{int(np.isnan(want).sum())}", flush=True)
