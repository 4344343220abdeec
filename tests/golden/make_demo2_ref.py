#!/usr/bin/env python3
"""Derive the statistical pin from the reference's only published output.

Input : /root/reference/demo.png -- demo2.yml at 16384 spp, 800x600 (README.md:1-3).
Output: tests/golden/demo2_ref_100x75.npy -- 8x8 box-filtered means, float32 [75][100][3] in [0,1].

The PNG carries the renderer's linear values (Image::write emits linear `c*65535.99`, image.rs:50-53;
the gAMA chunk is only a tag added by the PPM->PNG conversion), read here at 8 bits per channel.
This is data derived from a data file, not reference source.  Run once in the build container
(the GPU box has no /root/reference); the .npy is committed.
"""
import os
import numpy as np
from PIL import Image

here = os.path.dirname(os.path.abspath(__file__))
im = np.asarray(Image.open("/root/reference/demo.png").convert("RGB")).astype(np.float64) / 255.0
h, w, c = im.shape
assert (h, w, c) == (600, 800, 3)
small = im.reshape(h // 8, 8, w // 8, 8, c).mean(axis=(1, 3)).astype(np.float32)
np.save(os.path.join(here, "demo2_ref_100x75.npy"), small)
print(small.shape, small.mean(axis=(0, 1)))
