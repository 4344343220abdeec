#!/usr/bin/env python3
"""Regression fixtures rendered by the CPU oracle (SURVEY.md 8c "Regression fixtures"):
demo1 and demo2 geometry at 64x48 (pixel_size scaled to keep the field of view), 16 spp
(sample_root 4), depth 5, seed 1 -> float64 [48][64][3].  They pin CPU<->GPU agreement on the GPU
box and guard the oracle itself against accidental edits.  Not reference output (the reference is
non-deterministic and cannot be built here): these are regression data; the PIN to the reference is demo.png at
16 bits (make_demo2_ref16.py, oracle/flux_oracle.h).
"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import flux_amd  # noqa: E402
from conftest import small_scene  # noqa: E402
from oracle import oracle  # noqa: E402

for name in ("demo1", "demo2"):
    sd = small_scene(flux_amd.load_scene(os.path.join(root, "scenes", f"{name}.yml")), 64, 48)
    o = oracle.Oracle(sd, flux_amd.JobConfiguration(4, 5, 50), seed=1)
    img = o.render_frame(threads=1)
    np.save(os.path.join(here, f"{name}_64x48_n4_seed1.npy"), img)
    print(name, img.shape, img.mean())
