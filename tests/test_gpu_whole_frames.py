"""Whole frames of the PRODUCT build against the oracle, where the oracle costs seconds.

The GPU library has two instantiations of every render kernel: `<stats = true>` counts path statistics, `<stats = false>` is what
a render call without flux_ctx_enable_stats launches -- the product, and what bench.py times.  The statistics-equality tests
(tests/test_gpu_configs.py, tests/test_gpu_headline.py) necessarily run the first; here the frames of the SECOND are compared
directly, every row of BASELINE.json's configs 2 and 3 and every 25th row of the headline, in both arithmetics:

    config 2   scenes/demo1.yml 800x600 @  256 spp   600 rows   122.9 M samples on the CPU
    config 3   scenes/demo2.yml 800x600 @ 1024 spp   600 rows   491.5 M samples
    config 4   scenes/demo2.yml 800x600 @16384 spp    24 rows   314.6 M samples (rows 0, 25, ..., 575)

Camera::render (fluxcore/src/trace.rs:62-91) is what both sides compute; the bar is the north-star tolerance, 1e-4 per channel, on
every value, and 1e-9 at the 99.9th percentile (FP64 on both sides; the rare larger difference is a grazing ray whose hit / miss
decision flips between libm's and the device's last bits -- a single sample: 1/N of a pixel).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_IMAGE = 1e-4
TOL_TIGHT = 1e-9


def _threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 32))


def _compare(got, want, what):
    assert got.shape == want.shape
    assert np.isfinite(got).all(), what
    d = np.abs(got - want)
    assert d.max() < TOL_IMAGE, (what, float(d.max()))
    assert np.percentile(d, 99.9) < TOL_TIGHT, (what, float(np.percentile(d, 99.9)))
    return float(d.max())


def _frames(flux, sd, cfg, rows=None):
    """The product build's frame (or row list) in FAST and STRICT: statistics OFF, the default kernel."""
    out = {}
    with flux.Renderer(sd, cfg, seed=1) as r:
        for name, math in (("fast", flux.MATH_FAST), ("strict", flux.MATH_STRICT)):
            r.set_math(math)
            r.set_kernel(flux.KERNEL_DEFAULT)
            r.enable_stats(False)
            out[name] = r.render_frame() if rows is None else np.concatenate([r.render_rows(int(k), int(k)) for k in rows])
    return out


def test_config2_every_row_of_the_product_frame(flux, oracle_mod, demo1):
    cfg = flux.JobConfiguration(16, 5, 50)
    o = oracle_mod.Oracle(demo1, cfg, seed=1)
    want = o.render_frame(threads=_threads())
    o.close()
    got = _frames(flux, demo1, cfg)
    assert want.shape == (600, 800, 3)
    for name, img in got.items():
        print(f"config 2 {name}: max |gpu - oracle| over 600 rows = {_compare(img, want, name):.3e}")


def test_config3_every_row_of_the_product_frame(flux, oracle_mod, demo2):
    cfg = flux.JobConfiguration(32, 5, 50)
    o = oracle_mod.Oracle(demo2, cfg, seed=1)
    want = o.render_frame(threads=_threads())
    o.close()
    got = _frames(flux, demo2, cfg)
    for name, img in got.items():
        print(f"config 3 {name}: max |gpu - oracle| over 600 rows = {_compare(img, want, name):.3e}")


def test_headline_every_25th_row_of_the_product_frame(flux, oracle_mod, demo2):
    cfg = flux.JobConfiguration(128, 5, 50)
    rows = np.arange(0, 600, 25, dtype=np.int32)
    assert len(rows) == 24
    o = oracle_mod.Oracle(demo2, cfg, seed=1)
    want = o.render_row_list(rows, threads=min(_threads(), 24))
    o.close()
    got = _frames(flux, demo2, cfg, rows)
    for name, img in got.items():
        print(f"headline {name}: max |gpu - oracle| over 24 rows = {_compare(img, want, name):.3e}")
