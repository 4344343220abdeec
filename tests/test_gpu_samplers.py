"""The samplers crate's four generators on the device (flux_sampler_grid) against the oracle's restatement
(samplers/src/lib.rs:35-90,184-191), the stratification properties sampler-debug exists to eyeball
(sampler-debug/src/main.rs:25-57), and the sampler_debug tool's PPM output."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, max_abs_diff

pytestmark = pytest.mark.gpu

KINDS = {"r": 0, "j": 1, "mj": 2, "cmj": 3}


@pytest.mark.parametrize("n", [1, 2, 5, 10, 32])
def test_grids_match_oracle(flux, oracle_mod, n):
    seed = 11
    reg = flux.sampler_grid(KINDS["r"], n, seed)
    assert np.array_equal(reg, oracle_mod.grid_regular(n))
    key = oracle_mod.lib.fxo_rng_key(seed, 5, 0, 0, 0)
    assert np.array_equal(flux.sampler_grid(KINDS["j"], n, seed), oracle_mod.grid_jittered(key, n))
    # multi-jittered = hemi stream (kind 3) set 0 depth 0; correlated = pixel stream (kind 1) set 0
    assert np.array_equal(flux.sampler_grid(KINDS["mj"], n, seed), oracle_mod.grid_multi_jittered(seed, 3, 0, 0, n))
    assert np.array_equal(flux.sampler_grid(KINDS["cmj"], n, seed),
                          oracle_mod.grid_correlated_multi_jittered(seed, 1, 0, 0, n))


def test_hemisphere_images_and_properties(flux, oracle_mod):
    n, seed = 16, 3
    for name, kind in KINDS.items():
        xy, hm = flux.sampler_grid(kind, n, seed, hemi=True)
        assert xy.shape == (n * n, 2) and hm.shape == (n * n, 3)
        assert xy.min() >= 0.0 and xy.max() < 1.0
        want = np.array([oracle_mod.to_unit_hemi(x, y, 0.0) for x, y in xy])
        assert max_abs_diff(hm, want) < 1e-14  # sin/cos/pow: OCML vs glibc
        assert np.allclose(np.linalg.norm(hm, axis=1), 1.0, atol=1e-14) and hm[:, 2].min() > 0.0
        cells = set(zip((xy[:, 0] * n).astype(int), (xy[:, 1] * n).astype(int)))
        strips_x = set((xy[:, 0] * n * n).astype(int))
        strips_y = set((xy[:, 1] * n * n).astype(int))
        if name in ("r", "j", "cmj"):
            assert len(cells) == n * n          # one sample per coarse cell
        if name in ("mj", "cmj"):
            assert len(strips_x) == n * n and len(strips_y) == n * n  # n-rooks on the fine grid
    # the render tables are these generators: pixel_sets[0] is the correlated set
    sd = flux.load_scene(os.path.join(ROOT, "scenes", "demo1.yml"))
    with flux.Renderer(sd, flux.JobConfiguration(n, 2, 50), seed=seed) as r:
        assert np.array_equal(r.table(flux._lib.TABLE_PIXEL)[0], flux.sampler_grid(KINDS["cmj"], n, seed))
        assert max_abs_diff(r.table(flux._lib.TABLE_HEMI)[0, 0], flux.sampler_grid(KINDS["mj"], n, seed, hemi=True)[1]) == 0.0


def test_sampler_debug_tool(flux, tmp_path):
    from flux_amd import build
    build.build_host()
    exe = os.path.join(ROOT, "flux_amd", "host", "sampler_debug")
    out = subprocess.run([exe, "-r", "10", "--seed", "7", "--outdir", str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    for name, kind in KINDS.items():
        assert f"Wrote output to {tmp_path}/sampler-debug-{name}.ppm" in out.stdout
        toks = open(tmp_path / f"sampler-debug-{name}.ppm").read().split()
        assert toks[:4] == ["P3", "100", "100", "65535"]
        img = np.array(toks[4:], dtype=np.int64).reshape(100, 100, 3)
        xy, hm = flux.sampler_grid(kind, 10, 7, hemi=True)
        want = np.zeros((100, 100, 3), dtype=np.int64)
        for x, y in xy:  # plot_2d_sample (main.rs:12-16) + (c * 65535.99) as u16 (image.rs:49-52)
            want[int(y * 99.99), int(x * 99.99)] = (65535, 13107, 13107)
        assert np.array_equal(img, want) and (img[..., 0] > 0).sum() == 100
        hemi = np.array(open(tmp_path / f"sampler-debug-{name}-hemi.ppm").read().split()[4:], dtype=np.int64)
        hemi = hemi.reshape(100, 100, 3)
        want = np.zeros((100, 100, 3), dtype=np.int64)
        for x, y, z in hm:  # plot_hemi_sample (main.rs:18-23)
            want[int((y / 2.0 + 0.5) * 99.99), int((x / 2.0 + 0.5) * 99.99)] = (int(z * 65535.99), 13107, 13107)
        assert np.array_equal(hemi, want)
    bad = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert bad.returncode == 2
