"""Parity pin on the reference's ONLY published output at its real precision.

demo.png = scenes/demo2.yml rendered by the reference (README.md:1-3; written by fluxcore/src/image.rs:43-61 as a
16-bit P3 PPM), committed untouched as tests/golden/demo2_ref_800x600_u16.npy (tests/golden/make_demo2_ref16.py;
round 1 read it through PIL at 8 bits and box-filtered it).  The reference seeds its RNG from OS entropy
(samplers/src/lib.rs:27-33), so the image is one draw of the estimator, not a bitwise golden.  What CAN be pinned,
and is here, with the noise MEASURED from independent seeds of the renderer under test (tests/ref16.py):

 (a) first moment: whole-image, per-channel and per-region means agree within 1e-4 (the north-star tolerance) --
     measured 1e-5 on the whole image, i.e. the estimator's expectation is the reference's to 3e-5 relative;
 (b) per-pixel z-scores against the measured variance: unbiased, and no heavier-tailed than a held-out seed;
 (c) deterministic pixels (every sample returns the same radiance: glossy spheres mirroring the sky): equal to the
     reference within its 16-bit quantum;
 (d) second moment: the reference's per-pixel noise, region by region, is that of THIS estimator at sample_root 256;
     it is a quarter of the variance at the README's stated 16384 spp in every region (DESIGN.md section 2 discusses);
 (e) 8x8 block means (round 1's test) at what 16-bit data supports.
"""
import numpy as np
import pytest

import ref16

pytestmark = pytest.mark.gpu

M_SEEDS = 12          # independent FAST renders at 16384 spp that define mean and variance
HOLD_SEED = 101       # a 13th, used in the reference's place as the null case
TOL = 1e-4            # BASELINE.json north_star: per-channel tolerance


def _render(flux, sd, root, seed, math):
    with flux.Renderer(sd, flux.JobConfiguration(root, 5, 50), seed=seed) as r:
        r.set_math(math)
        return r.render_frame()


@pytest.fixture(scope="module")
def draws(flux, demo2):
    """M_SEEDS + 1 renders at sample_root 128 and 2 at sample_root 256 (FAST): ~15 s of GPU time, shared."""
    frames = [_render(flux, demo2, 128, s, flux.MATH_FAST) for s in range(2, M_SEEDS + 2)]
    mean, var = ref16.seed_moments(frames)
    return {
        "frames": frames, "mean": mean, "var": var,
        "hold": _render(flux, demo2, 128, HOLD_SEED, flux.MATH_FAST),
        "r256": [_render(flux, demo2, 256, s, flux.MATH_FAST) for s in (HOLD_SEED, HOLD_SEED + 1)],
        "ref": ref16.load_ref16(), "omap": ref16.object_map(demo2),
    }


def _regions(omap):
    from scipy.ndimage import binary_erosion
    out = {"whole": np.ones(omap.shape, dtype=bool)}
    for k in np.unique(omap):
        out["floor" if k == 12 else f"sphere{k}"] = binary_erosion(omap == k, iterations=3)
    h, w = omap.shape
    for name, sl in (("top-left", (slice(0, h // 2), slice(0, w // 2))), ("top-right", (slice(0, h // 2), slice(w // 2, w))),
                     ("bottom-left", (slice(h // 2, h), slice(0, w // 2))), ("bottom-right", (slice(h // 2, h), slice(w // 2, w)))):
        m = np.zeros(omap.shape, dtype=bool)
        m[sl] = True
        out[name] = m
    return out


def _check_means(ref, mean, frames, omap, label):
    """(a): region means of the reference against the seed-mean; the seed-to-seed spread of the same region mean
    gives the standard error without assuming independent pixels (pixels sharing a sample set are correlated)."""
    report = {}
    for name, m in _regions(omap).items():
        if m.sum() < 500:
            continue
        d = (ref - mean)[m].mean(axis=0)
        per_seed = np.array([f[m].mean(axis=0) for f in frames])
        se1 = per_seed.std(axis=0, ddof=1)                      # one 16384-spp draw
        se = se1 * np.sqrt(1.0 + 1.0 / len(frames))
        report[name] = (d, se)
        bound = np.maximum(TOL, 5.0 * se)
        assert np.all(np.abs(d) < bound), f"{label} {name}: mean diff {d} exceeds {bound} (se {se})"
    d, _ = report["whole"]
    assert np.all(np.abs(d) < TOL) and abs(d.mean()) < TOL, f"{label}: whole-image mean diff {d}"
    return report


def test_reference_means_fast(draws):
    rep = _check_means(draws["ref"], draws["mean"], draws["frames"], draws["omap"], "FAST")
    d, se = rep["whole"]
    print(f"whole-image mean diff (reference - GPU mean of {M_SEEDS} seeds): {d}, one-draw se {se}")
    # the floor (68 % of the image, Matte + the uniform-hemisphere quirk) and the largest spheres at the tolerance
    for name in ("floor", "sphere2", "sphere3", "sphere4"):
        assert np.all(np.abs(rep[name][0]) < TOL), (name, rep[name])


def test_reference_means_strict(flux, demo2, draws):
    """The reference-order arithmetic through the same pin (4 seeds; STRICT renders at a third of FAST's rate)."""
    frames = [_render(flux, demo2, 128, s, flux.MATH_STRICT) for s in range(2, 6)]
    # same seeds => same samples: STRICT and FAST frames differ by rounding only
    assert max(float(np.abs(a - b).max()) for a, b in zip(frames, draws["frames"][:4])) < 1e-9
    _check_means(draws["ref"], np.mean(frames, axis=0), frames, draws["omap"], "STRICT")


def test_per_pixel_zscores(draws):
    """(b): z = (reference - mean) / sqrt(var (1 + 1/M) + quantisation) over the pixels whose estimator has variance.
    The held-out seed calibrates the statistic (it IS a draw of the estimator at 16384 spp): the reference must be
    unbiased and no more dispersed / heavier-tailed than it."""
    z_ref, noisy = ref16.zscores(draws["ref"], draws["mean"], draws["var"], M_SEEDS)
    z_hold, _ = ref16.zscores(draws["hold"], draws["mean"], draws["var"], M_SEEDS)
    zr, zh = z_ref[noisy], z_hold[noisy]
    assert noisy.mean() > 0.8
    stats = {k: (float(np.mean(z)), float(np.std(z)), float((np.abs(z) > 3).mean()), float((np.abs(z) > 6).mean()))
             for k, z in (("ref", zr), ("hold", zh))}
    print("z-scores (mean, std, frac |z|>3, frac |z|>6):", stats)
    assert abs(stats["hold"][0]) < 0.03 and 0.9 < stats["hold"][1] < 1.4      # the null case behaves
    assert abs(stats["ref"][0]) < 0.03                                       # no per-pixel bias
    assert stats["ref"][1] < stats["hold"][1] and stats["ref"][1] > 0.4       # noise: less than one 16384-spp draw
    assert stats["ref"][2] <= stats["hold"][2] and stats["ref"][3] <= max(stats["hold"][3], 2e-4)


def test_deterministic_pixels(draws):
    """(c): pixel channels with zero seed-to-seed variance are deterministic -- every one of the 16384 paths returns
    the same radiance (a glossy sphere mirroring only the environment emitter: cs ks * power * colour, or a clamped
    highlight).  There the reference must agree to its own 16-bit quantum."""
    quiet = draws["var"] < 1e-20
    assert quiet.mean() > 0.05                       # ~13 % of the image
    d = np.abs(draws["ref"] - draws["mean"])[quiet] * ref16.PPM_SCALE
    print("deterministic pixel-channels:", int(quiet.sum()), "median |d| in quanta", float(np.median(d)),
          "frac > 1 quantum", float((d > 1.0).mean()))
    assert np.median(d) <= 0.5                        # inside the quantisation cell
    assert (d > 1.0).mean() < 3e-3                   # the rest: clamp-edge pixels where a rare path crosses 1.0
    # and the value is the closed form: exponent-1e4 spheres mirror the sky as 0.5*(0.8,0.6,1.0) * 0.3*(1,0.9686,0.8588)
    expect = 0.5 * np.array([0.8, 0.6, 1.0]) * 0.3 * np.array([1.0, 0.9686, 0.8588])
    px = np.all(quiet, axis=2) & np.all(np.abs(draws["mean"] - expect) < 1e-12, axis=2)
    assert px.sum() > 100
    dq = np.abs(draws["ref"][px] - expect) * ref16.PPM_SCALE
    assert np.median(dq) <= 0.5 and (dq > 1.0).mean() < 5e-3   # a rare path of the reference's own draw may stray


def test_variance_profile(draws):
    """(d): the reference's per-pixel noise.  v_x = E[(x - mean)^2] - v/M per region, in units of the estimator's
    variance v at 16384 spp: 1.0 for a held-out seed (the null case), ~0.25-0.5 for the reference -- the same values,
    region by region, as this estimator at sample_root 256 (65536 spp)."""
    mean, var = draws["mean"], draws["var"]

    def ratio(x, m):
        v = var[m].mean()
        return (((x - mean)[m] ** 2).mean() - v / M_SEEDS) / v

    rows = {}
    for name, m in _regions(draws["omap"]).items():
        if m.sum() < 2000 or name.startswith(("top", "bottom")):
            continue
        rows[name] = (ratio(draws["ref"], m), ratio(draws["hold"], m), np.mean([ratio(x, m) for x in draws["r256"]]))
    print("region: reference / held-out 16384 spp / 65536 spp:", {k: tuple(round(float(x), 3) for x in v) for k, v in rows.items()})
    for name, (r_ref, r_hold, r_256) in rows.items():
        assert 0.75 < r_hold < 1.3, (name, r_hold)                    # the statistic is calibrated
        assert 0.6 * r_256 < r_ref < 1.7 * r_256, (name, r_ref, r_256)  # heavy-tailed on the small spheres
    for name in ("whole", "floor"):
        r_ref, _, r_256 = rows[name]
        assert abs(r_ref / r_256 - 1.0) < 0.15, (name, r_ref, r_256)
        assert 0.2 < r_ref < 0.4                                      # a quarter of the 16384-spp variance


def test_block_means(draws):
    """(e): round 1's 8x8 block-mean comparison, now against the 16-bit data and the mean of M seeds (round 1: one seed
    against the 8-bit file, mean |d| < 0.0025): measured mean 2.0e-4, 99th percentile 1.2e-3, max 4.3e-3."""
    d8 = (draws["ref"] - draws["mean"]).reshape(75, 8, 100, 8, 3).mean(axis=(1, 3))
    print("8x8 blocks: mean |d|", float(np.abs(d8).mean()), "p99", float(np.percentile(np.abs(d8), 99)), "max",
          float(np.abs(d8).max()))
    assert np.abs(d8).mean() < 3.5e-4
    assert np.percentile(np.abs(d8), 99) < 2.5e-3 and np.abs(d8).max() < 8e-3
    assert np.all(np.abs(d8.mean(axis=(0, 1))) < TOL)
    h8 = (draws["hold"] - draws["mean"]).reshape(75, 8, 100, 8, 3).mean(axis=(1, 3))
    assert np.abs(d8).mean() < np.abs(h8).mean()      # closer to the mean than one 16384-spp draw is
