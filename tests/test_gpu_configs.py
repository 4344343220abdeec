"""The BASELINE.json configurations at FULL size, through size-independent properties (the oracle does not finish
them in seconds), plus the pieces that only exist for them:

 * config 2  scenes/demo1.yml 800x600 @256 spp (sample_root 16): determinism, kernel variants equal up to summation
             order, finite / [0,1], path-statistics identities and their equality across kernels; rows 0, 1, 299, 300 of the
             full-size frame against the oracle (1e-4 / 1e-9 at the 99.9th percentile, equal statistics), FAST and STRICT;
 * config 3  scenes/demo2.yml 800x600 @1024 spp (sample_root 32), the FULL frame: determinism, split == refill == static up to
             summation order with identical path statistics, finite / [0,1], rows 0-1 and 298-301 against the oracle;
 * config 4  the per-rank contexts of the set-sharded frame (flux_ctx_create_sets): same tables, same pixels as the
             full context; and the "misses" of an ENCLOSED scene -- rays that start within T_MIN of the environment
             sphere (scene.rs:156-160 + constants.rs:4: the hit is rejected) -- agree with the oracle ray by ray;
 * config 5  the procedural 1M-triangle height field (extension; flux_amd/procedural.py): BVH traversal == brute force
             over all 1,000,000 triangles on thousands of rays in both arithmetics, identical path statistics on a
             pixel window, the same horizon-miss behaviour, and ONE FULL 4096-spp FRAME through render_bvh4_kernel
             (determinism, statistics identities, finite / [0,1], rows 330-331 against the static kernel).
"""
import numpy as np
import pytest

from conftest import max_abs_diff

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ config 2
@pytest.mark.parametrize("math", ["fast", "strict"])
def test_config2_full_size_properties(flux, oracle_mod, demo1, math):
    mode = flux.MATH_FAST if math == "fast" else flux.MATH_STRICT
    with flux.Renderer(demo1, flux.JobConfiguration(16, 5, 50), seed=1) as r:
        r.set_math(mode)
        r.enable_stats(True)
        frames, stats = {}, {}
        for name, variant in (("default", flux.KERNEL_DEFAULT), ("static", flux.KERNEL_STATIC),
                              ("refill", flux.KERNEL_REFILL), ("split", flux.KERNEL_SPLIT)):
            r.set_kernel(variant)
            r.stats(reset=True)
            frames[name] = r.render_frame()
            stats[name] = r.stats(reset=True)
        r.set_kernel(flux.KERNEL_DEFAULT)
        again = r.render_frame()
    a = frames["default"]
    assert a.shape == (600, 800, 3)
    assert np.array_equal(a, again)                                     # bitwise run-to-run determinism
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0   # max_to_one (color.rs:35-44)
    for name in ("static", "refill", "split"):
        assert max_abs_diff(a, frames[name]) < 1e-12, name              # same samples, another fixed summation order
        assert stats[name] == stats["default"], name                    # every hit / material decision identical
    st = stats["default"]
    assert st["samples"] == 800 * 600 * 256
    assert st["segments"] == st["matte_bounces"] + st["glossy_bounces"] + st["specular_bounces"] + \
        st["emissive_hits"] + st["misses"]
    assert st["samples"] == st["emissive_hits"] + st["misses"] + st["depth_exhausted"]  # every path ends exactly once
    assert st["specular_bounces"] == 0 and st["matte_bounces"] > 0 and st["glossy_bounces"] > 0
    # the work-unit decomposition (Job::work_units, job.rs:65-88) reproduces the frame bit for bit
    with flux.Renderer(demo1, flux.JobConfiguration(16, 5, 50), seed=1) as r:
        r.set_math(mode)
        for u in flux.work_units(600, 50)[::5]:
            assert np.array_equal(r.render_rows(u.row_start, u.row_end), a[u.row_start:u.row_end + 1])
        # ... and four rows of the FULL-SIZE frame meet the oracle (trace.rs:71-87: 4 x 800 x 256 = 0.8 M samples on the CPU): the
        # horizon rows 0-1 and rows 299-300 through the spheres, at the north-star tolerance, with equal path statistics
        rows = np.array([0, 1, 299, 300], dtype=np.int32)
        o = oracle_mod.Oracle(demo1, flux.JobConfiguration(16, 5, 50), seed=1)
        want = o.render_row_list(rows)
        assert max_abs_diff(a[rows], want) < 1e-4
        assert np.percentile(np.abs(a[rows] - want), 99.9) < 1e-9
        r.enable_stats(True)
        for row in (0, 300):
            r.stats(reset=True)
            r.render_rows(row, row)
            got = r.stats(reset=True)
            o.stats(reset=True)
            o.render_row_list(np.array([row], dtype=np.int32))
            ost = o.stats(reset=True)
            for k in ("samples", "segments", "matte_bounces", "glossy_bounces", "specular_bounces", "emissive_hits", "misses", "depth_exhausted"):
                assert got[k] == ost[k], (row, k, got[k], ost[k])
        o.close()


# ------------------------------------------------------------------------------------------------ config 3
def test_config3_full_frame(flux, oracle_mod, demo2):
    """BASELINE config 3 as stated: scenes/demo2.yml, the whole 800x600 frame at 1024 spp (Camera::render over all rows,
    trace.rs:62-91).  Size-independent properties on the full frame, and six rows -- the two at the top (the horizon) and
    four through the middle of the spheres -- against the oracle at the north-star tolerance."""
    cfg = flux.JobConfiguration(32, 5, 50)
    with flux.Renderer(demo2, cfg, seed=1) as r:
        r.enable_stats(True)
        frames, stats = {}, {}
        for name, variant in (("split", flux.KERNEL_SPLIT), ("refill", flux.KERNEL_REFILL), ("static", flux.KERNEL_STATIC)):
            r.set_kernel(variant)
            r.stats(reset=True)
            frames[name] = r.render_frame()
            stats[name] = r.stats(reset=True)
        r.set_kernel(flux.KERNEL_DEFAULT)
        r.enable_stats(False)
        again = r.render_frame()
    a = frames["split"]
    assert a.shape == (600, 800, 3)
    assert np.array_equal(a, again)                                     # the default kernel IS the split kernel; bitwise repeatable
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0   # max_to_one (color.rs:35-44)
    for name in ("refill", "static"):
        assert max_abs_diff(a, frames[name]) < 1e-12, name              # same samples, another fixed summation order
        assert stats[name] == stats["split"], name                      # every hit / material decision identical
    st = stats["split"]
    assert st["samples"] == 800 * 600 * 1024
    assert st["segments"] == st["matte_bounces"] + st["glossy_bounces"] + st["specular_bounces"] + st["emissive_hits"] + st["misses"]
    assert st["samples"] == st["emissive_hits"] + st["misses"] + st["depth_exhausted"]
    rows = np.array([0, 1, 298, 299, 300, 301], dtype=np.int32)
    want = oracle_mod.Oracle(demo2, cfg, seed=1).render_row_list(rows)   # 6 x 800 x 1024 = 4.9 M samples on the CPU
    assert max_abs_diff(a[rows], want) < 1e-4
    assert np.percentile(np.abs(a[rows] - want), 99.9) < 1e-9


# ------------------------------------------------------------------------------------------------ config 4
@pytest.mark.parametrize("world", [1, 3, 8])
def test_set_share_context_equals_full_context(flux, demo2, small, world):
    """A rank's context (flux_ctx_create_sets: tables for the sets rank + k*world only) holds the same table contents
    and renders the same pixels as the full context's share; rendering rows on it is refused."""
    import torch
    sd = small(demo2, 40, 24)
    cfg = flux.JobConfiguration(8, 5, 50)
    dev = torch.device("cuda", 0)
    with flux.Renderer(sd, cfg, seed=3) as full:
        pix, disc, hemi = (full.table(w) for w in (flux._lib.TABLE_PIXEL, flux._lib.TABLE_DISC, flux._lib.TABLE_HEMI))
        for rank in range(world):
            count = len(range(rank, 40, world))
            want = torch.zeros((24, count, 3), dtype=torch.float64, device=dev)
            full.render_sets_device(rank, world, count, want.data_ptr())
            with flux.Renderer(sd, cfg, seed=3, set_share=(rank, world)) as part:
                assert np.array_equal(part.table(flux._lib.TABLE_PIXEL), pix[rank::world])
                assert np.array_equal(part.table(flux._lib.TABLE_DISC), disc[rank::world])
                assert np.array_equal(part.table(flux._lib.TABLE_HEMI), hemi[rank::world])
                assert part.device_bytes() < full.device_bytes() or world == 1
                got = torch.zeros_like(want)
                part.render_sets_device(rank, world, count, got.data_ptr())
                torch.cuda.synchronize()
                assert torch.equal(got, want)
                if world > 1:
                    with pytest.raises(flux.FluxError):
                        part.render_rows(0, 0)                               # a row needs every set
                    with pytest.raises(flux.FluxError):
                        part.render_sets_device((rank + 1) % world, world, 1, got.data_ptr())  # not this rank's sets
            torch.cuda.synchronize()
    with pytest.raises(flux.FluxError):
        flux.Renderer(sd, cfg, seed=3, set_share=(3, 3))                     # first_set < set_stride
    with pytest.raises(flux.FluxError):
        flux.Renderer(sd, cfg, seed=3, set_share=(0, 1 << 32))               # the stride is held in 32 bits
    # more ranks than sample sets: rank 45 of 48 owns no set of this 40-set image -- an EMPTY share is a valid context
    # (no per-set tables, the row permutations only) whose render is a no-op
    with flux.Renderer(sd, cfg, seed=3, set_share=(45, 48)) as empty, flux.Renderer(sd, cfg, seed=3) as full:
        assert empty.table(flux._lib.TABLE_PIXEL).shape[0] == 0
        assert np.array_equal(empty.row_perm(5), full.row_perm(5))
        empty.render_sets_device(45, 48, 0, 0)
        assert empty.device_bytes() < full.device_bytes()


def _horizon_rays(rng, n, radius, y_plane):
    """Origins on the ground plane within 2 T_MIN of the environment sphere (inside it), directions into the upper
    hemisphere, half of them roughly outward: the sphere's exit distance straddles T_MIN = 0.0005."""
    phi = rng.uniform(0, 2 * np.pi, n)
    rho2 = radius * radius - y_plane * y_plane
    rho = np.sqrt(rho2) - rng.uniform(0.0, 0.001, n)
    o = np.stack([rho * np.cos(phi), np.full(n, y_plane), rho * np.sin(phi)], axis=1)
    d = rng.normal(size=(n, 3))
    d[:, 1] = np.abs(d[:, 1]) * rng.uniform(0.0, 1.0, n)
    outward = rng.uniform(size=n) < 0.5
    d[outward, 0] = np.abs(d[outward, 0]) * np.sign(o[outward, 0]) * 3
    d[outward, 2] = np.abs(d[outward, 2]) * np.sign(o[outward, 2]) * 3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


@pytest.mark.parametrize("math", ["fast", "strict"])
def test_enclosed_scene_misses_are_the_oracles(flux, oracle_mod, demo2, math):
    """bench.py reports `misses: 25` for demo2 @16384 spp although the inverted environment sphere encloses the scene:
    paths that reach the ground plane within T_MIN of the sphere and leave outward -- the sphere's hit distance is below
    T_MIN and the reference rejects it (shapes.rs:189-205), so Scene::hit returns None and the background is used
    (scene.rs:168).  Reproduced ray by ray against the oracle."""
    rng = np.random.default_rng(77)
    o, d = _horizon_rays(rng, 4096, 100.0, 0.0)
    cfg = flux.JobConfiguration(2, 5, 50)
    orc = oracle_mod.Oracle(demo2, cfg, seed=1)
    want = np.array([orc.scene_hit(o[k], d[k])[0] for k in range(len(o))])
    with flux.Renderer(demo2, cfg, seed=1) as r:
        r.set_math(flux.MATH_FAST if math == "fast" else flux.MATH_STRICT)
        _, hit, _ = r.debug_shade(o, d, depth=5)
    assert np.array_equal(hit, want)
    assert 50 < (want == -1).sum() < 3000        # a good share of them ARE misses, and not all
    # and at the frame level: the full 800x600 frame at 64 spp has the same statistics in every kernel
    # (misses included), and rows 0..1 -- where the horizon is -- equal the oracle's
    with flux.Renderer(demo2, flux.JobConfiguration(8, 5, 50), seed=1) as r:
        r.set_math(flux.MATH_FAST if math == "fast" else flux.MATH_STRICT)
        r.enable_stats(True)
        per = {}
        for variant in (flux.KERNEL_STATIC, flux.KERNEL_REFILL, flux.KERNEL_SPLIT):
            r.set_kernel(variant)
            r.stats(reset=True)
            r.render_frame()
            per[variant] = r.stats(reset=True)
        assert per[flux.KERNEL_STATIC] == per[flux.KERNEL_REFILL] == per[flux.KERNEL_SPLIT]


# ------------------------------------------------------------------------------------------------ config 5
@pytest.fixture(scope="module")
def hf_scene():
    from flux_amd.procedural import heightfield_scene
    return heightfield_scene(1000, 500)


@pytest.fixture(scope="module")
def hf_renderer(flux, hf_scene):
    r = flux.Renderer(hf_scene, flux.JobConfiguration(2, 5, 50), seed=1)
    info = r.bvh_info()
    assert info["triangles"] == 1_000_000 and info["max_depth"] <= 64
    yield r
    r.close()


def _hf_rays(rng, n):
    """Primary-like rays from the camera, bounce-like rays leaving the height field in all upward directions, grazing
    rays along the field, and rays from far outside the mesh's box."""
    k = n // 4
    eye = np.array([0.0, 5.5, -9.0])
    tgt = np.stack([rng.uniform(-14, 14, k), rng.uniform(-0.4, 0.4, k), rng.uniform(-10, 20, k)], axis=1)
    o1, d1 = np.tile(eye, (k, 1)), tgt - eye
    o2 = np.stack([rng.uniform(-13.9, 13.9, k), rng.uniform(-0.3, 0.6, k), rng.uniform(-9.9, 19.9, k)], axis=1)
    d2 = rng.normal(size=(k, 3))
    o3 = np.stack([rng.uniform(-14, 14, k), rng.uniform(-0.35, 0.35, k), rng.uniform(-10, 20, k)], axis=1)
    d3 = rng.normal(size=(k, 3))
    d3[:, 1] *= 0.02                                                   # grazing: long walks through the BVH
    o4 = rng.uniform(-60, 60, (n - 3 * k, 3))
    d4 = np.stack([rng.uniform(-14, 14, n - 3 * k), rng.uniform(-0.4, 0.4, n - 3 * k), rng.uniform(-10, 20, n - 3 * k)], axis=1) - o4
    o = np.concatenate([o1, o2, o3, o4])
    d = np.concatenate([d1, d2, d3, d4])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a few axis-aligned directions (zero components: the slab test's 1/0 handling)
    d[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    return o, d


@pytest.mark.parametrize("math,n_rays", [("fast", 16384), ("strict", 4096)])
def test_config5_bvh_equals_bruteforce_on_1m_triangles(flux, hf_renderer, math, n_rays):
    """First hit (id exact, distance to 1e-12) and one Scene::shade level of BVH traversal against the brute-force scan
    of all 1,000,000 triangles -- the definition the BVH must reproduce: builder depth, LDS stack
    bound and the conservative f32 slabs at this size."""
    r = hf_renderer
    r.set_math(flux.MATH_FAST if math == "fast" else flux.MATH_STRICT)
    o, d = _hf_rays(np.random.default_rng(5), n_rays)
    r.set_traversal(flux._lib.TRAVERSE_BVH)
    rgb_b, hit_b, t_b = r.debug_shade(o, d, depth=5, set_index=3, sample_index=1)
    r.set_traversal(flux._lib.TRAVERSE_BRUTE)
    rgb_f, hit_f, t_f = r.debug_shade(o, d, depth=5, set_index=3, sample_index=1)
    r.set_traversal(flux._lib.TRAVERSE_BVH)
    assert np.array_equal(hit_b, hit_f)
    tri = hit_f >= 13                                   # 12 spheres + the plane precede the triangles in hit order
    assert tri.sum() > n_rays // 4 and (~tri).sum() > n_rays // 20
    assert np.abs(t_b - t_f).max() <= 1e-12 * max(1.0, np.abs(t_f).max())
    assert max_abs_diff(rgb_b, rgb_f) <= 1e-12


def test_config5_pixel_window_statistics(flux, hf_renderer):
    """A pixel window of the real frame (rows 330-331: the field's far edge and the spheres, 4 spp): BVH render ==
    brute-force render -- identical path statistics (every segment's hit decision) and, in the static kernel whose
    summation order is shared, the identical image; the traversal state-machine kernel to 1e-12."""
    r = hf_renderer
    r.set_math(flux.MATH_FAST)
    r.enable_stats(True)
    out = {}
    for trav in (flux._lib.TRAVERSE_BVH, flux._lib.TRAVERSE_BRUTE):
        r.set_traversal(trav)
        r.set_kernel(flux.KERNEL_STATIC)
        r.stats(reset=True)
        img = r.render_rows(330, 331)
        st = r.stats(reset=True)
        out[trav] = (img, {k: v for k, v in st.items() if k not in ("bvh_nodes", "tris_tested")}, st)
    r.set_traversal(flux._lib.TRAVERSE_BVH)
    r.set_kernel(flux.KERNEL_DEFAULT)
    r.enable_stats(False)
    (img_b, st_b, raw_b), (img_f, st_f, raw_f) = out[flux._lib.TRAVERSE_BVH], out[flux._lib.TRAVERSE_BRUTE]
    assert st_b == st_f and st_b["samples"] == 2 * 800 * 4
    assert np.array_equal(img_b, img_f)
    assert raw_f["tris_tested"] == raw_f["segments"] * 1_000_000 and raw_b["tris_tested"] < raw_f["tris_tested"] // 10_000


def test_config5_state_machine_kernel_and_misses(flux, oracle_mod, hf_scene):
    """The default kernel for this configuration (render_bvh_kernel: persistent lanes with a traversal state machine,
    >= 64 spp) against brute force on a row of the frame, path statistics included; and the horizon rays -- the source
    of the `misses` bench.py reports for this enclosed scene -- against brute force and against the oracle's
    1,000,000-triangle scan."""
    with flux.Renderer(hf_scene, flux.JobConfiguration(8, 5, 50), seed=1) as r:
        r.enable_stats(True)
        res = {}
        for trav in (flux._lib.TRAVERSE_BVH, flux._lib.TRAVERSE_BRUTE):
            r.set_traversal(trav)
            r.stats(reset=True)
            res[trav] = (r.render_rows(331, 331), r.stats(reset=True))
        (img_b, st_b), (img_f, st_f) = res[flux._lib.TRAVERSE_BVH], res[flux._lib.TRAVERSE_BRUTE]
        drop = ("bvh_nodes", "tris_tested")
        assert {k: v for k, v in st_b.items() if k not in drop} == {k: v for k, v in st_f.items() if k not in drop}
        assert max_abs_diff(img_b, img_f) < 1e-12
        # horizon rays on the y = -1 catch plane, within T_MIN of the environment sphere
        o, d = _horizon_rays(np.random.default_rng(78), 2048, 100.0, -1.0)
        r.set_traversal(flux._lib.TRAVERSE_BVH)
        _, hit_b, _ = r.debug_shade(o, d, depth=5)
        r.set_traversal(flux._lib.TRAVERSE_BRUTE)
        _, hit_f, _ = r.debug_shade(o, d, depth=5)
        assert np.array_equal(hit_b, hit_f)
        assert 20 < (hit_b == -1).sum() < 1500
    orc = oracle_mod.Oracle(hf_scene, flux.JobConfiguration(2, 5, 50), seed=1)
    sub = np.concatenate([np.flatnonzero(hit_b == -1)[:24], np.flatnonzero(hit_b != -1)[:24]])
    want = np.array([orc.scene_hit(o[k], d[k])[0] for k in sub])      # ~5 ms per ray on the CPU: a sample
    assert np.array_equal(hit_b[sub], want)


def test_config5_full_frame_at_4096_spp(flux, hf_scene):
    """BASELINE config 5 as stated, on the one GPU a test box has: the whole 800x600 frame of the 1,000,000-triangle scene at
    4096 spp through render_bvh4_kernel, the 4-wide-tree state machine (1.97 G camera paths).  No CPU comparison is possible at this size (the oracle scans
    every triangle per ray), so: bitwise determinism, the statistics identities, finite / [0,1]; and rows 330-331 (the field's
    far edge and the spheres) at the same 4096 spp against the STATIC kernel, whose inline BVH walk shares nothing with the
    state machine but the hit rule -- equal statistics, images to summation order (the static kernel itself equals brute
    force bit for bit: test_config5_pixel_window_statistics)."""
    with flux.Renderer(hf_scene, flux.JobConfiguration(64, 5, 50), seed=1) as r:
        a = r.render_frame()
        b = r.render_frame()
        assert np.array_equal(a, b)
        assert a.shape == (600, 800, 3) and np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0
        r.enable_stats(True)
        r.stats(reset=True)
        c = r.render_frame()
        st = r.stats(reset=True)
        assert np.array_equal(a, c)                                     # counting statistics changes no pixel
        assert st["samples"] == 800 * 600 * 4096
        assert st["segments"] == st["matte_bounces"] + st["glossy_bounces"] + st["specular_bounces"] + st["emissive_hits"] + st["misses"]
        assert st["samples"] == st["emissive_hits"] + st["misses"] + st["depth_exhausted"]
        assert st["bvh_nodes"] > 10 * st["segments"] and st["tris_tested"] > st["segments"]
        band, stb = r.render_rows(330, 331), r.stats(reset=True)
        r.set_kernel(flux.KERNEL_STATIC)
        ref, sts = r.render_rows(330, 331), r.stats(reset=True)
        drop = ("bvh_nodes", "tris_tested")                             # the two kernels order their traversals differently
        assert {k: v for k, v in stb.items() if k not in drop} == {k: v for k, v in sts.items() if k not in drop}
        assert np.array_equal(band, a[330:332])
        assert max_abs_diff(band, ref) < 1e-12
