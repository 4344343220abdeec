"""Shared helpers for the parity pin on the reference's one published render at its real 16-bit precision.

tests/golden/demo2_ref_800x600_u16.npy holds demo.png's samples (tests/golden/make_demo2_ref16.py): demo2.yml at
16384 spp, 800x600 (README.md:1-3), each sample v = `(c * 65535.99) as u16` (fluxcore/src/image.rs:50-53) of the
averaged, max_to_one-clamped pixel colour c.  The reference seeds its RNG from OS entropy
(samplers/src/lib.rs:27-33), so the comparison is between two draws of the same estimator: the noise model is
measured, not assumed -- the per-pixel variance of the estimator comes from M independent seeds of the SAME
renderer, and every statistic is also evaluated for a held-out seed in the reference's place (the null case).
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PPM_SCALE = 65535.99  # image.rs:50-53


def load_ref16():
    """Reference pixel colours, [600][800][3] f64: the centre of each quantisation cell [v, v+1) / 65535.99."""
    v = np.load(os.path.join(GOLDEN, "demo2_ref_800x600_u16.npy"))
    assert v.shape == (600, 800, 3) and v.dtype == np.uint16
    return (v.astype(np.float64) + 0.5) / PPM_SCALE


QUANT_VAR = (1.0 / PPM_SCALE) ** 2 / 12.0  # variance of the uniform quantisation error


def object_map(sd):
    """First shape (YAML index) seen through the CENTRE of each pixel by a pinhole ray from the eye
    (trace.rs:44-51,72-80 with lens sample 0 and pixel sample (0.5,0.5)); -1 = nothing.  Only used to cut the image
    into regions (sky / light / each sphere / floor); both renders are blurred by the same lens."""
    from flux_amd.scene import PlaneData, SphereData
    W, H = sd.output_settings.image_width, sd.output_settings.image_height
    ps = sd.output_settings.pixel_size / sd.camera_data.zoom_factor
    eye = np.array(sd.camera_settings.eye, dtype=np.float64)
    look = np.array(sd.camera_settings.look_at, dtype=np.float64)
    up = np.array(sd.camera_settings.up, dtype=np.float64)
    w = eye - look
    w /= np.linalg.norm(w)
    u = np.cross(up, w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    cols = np.arange(W)[None, :] - W / 2 + 0.5
    rows = (H - np.arange(H))[:, None] - H / 2 + 0.5
    k = sd.camera_data.focal_distance / sd.camera_data.view_plane_distance
    a = ps * cols * k
    b = ps * rows * k
    d = a[..., None] * u + b[..., None] * v - sd.camera_data.focal_distance * w
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    best_t = np.full((H, W), np.inf)
    best = np.full((H, W), -1, dtype=np.int32)
    for idx, s in enumerate(sd.shapes):
        if isinstance(s, SphereData):
            oc = eye - np.array(s.center, dtype=np.float64)
            hb = d @ oc
            c = oc @ oc - s.radius * s.radius
            disc = hb * hb - c
            ok = disc >= 0
            e = np.sqrt(np.where(ok, disc, 0.0))
            t = np.where(-hb - e > 1e-5, -hb - e, -hb + e)
            ok &= t > 1e-5
        elif isinstance(s, PlaneData):
            n = np.array(s.normal, dtype=np.float64)
            den = d @ n
            with np.errstate(divide="ignore", invalid="ignore"):
                t = ((np.array(s.point, dtype=np.float64) - eye) @ n) / den
            ok = t > 1e-5
        else:
            continue
        take = ok & (t < best_t)
        best_t = np.where(take, t, best_t)
        best = np.where(take, idx, best)
    return best


def seed_moments(frames):
    """Mean and unbiased per-pixel variance of M independent renders (the estimator's variance at this spp)."""
    f = np.asarray(frames, dtype=np.float64)
    return f.mean(axis=0), f.var(axis=0, ddof=1)


def aggregate_stats(ref, mean, var, m, mask=None):
    """Difference of region means between a single draw `ref` and the mean of m seeds, with the standard error the
    measured per-pixel variance predicts for it (pixels treated as independent; channels separately)."""
    if mask is None:
        mask = np.ones(ref.shape[:2], dtype=bool)
    d = (ref - mean)[mask]          # [n][3]
    v = var[mask] * (1.0 + 1.0 / m) + QUANT_VAR
    n = d.shape[0]
    diff = d.mean(axis=0)
    se = np.sqrt(v.sum(axis=0)) / n
    return diff, se, n


def zscores(ref, mean, var, m, floor=1e-12):
    """Per-pixel, per-channel z = (ref - mean) / sqrt(var (1 + 1/m) + quantisation); pixels whose estimator has no
    variance (directly seen emitters: every sample returns the same colour) are returned separately."""
    s2 = var * (1.0 + 1.0 / m) + QUANT_VAR
    noisy = var > floor
    z = (ref - mean) / np.sqrt(s2)
    return z, noisy
