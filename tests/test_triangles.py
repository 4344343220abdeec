"""Extension: triangle meshes + BVH (absent in the reference: scene.rs:71-74 is Sphere | Plane only, so there
is no reference oracle).  The definition is the brute-force scan in the CPU oracle (Moeller-Trumbore f64,
Plane's conventions); the GPU's BVH traversal must return exactly the brute-force nearest hit."""
import copy

import numpy as np
import pytest

from conftest import max_abs_diff, small_scene


def _tri_scene(flux, demo2, meshes, w=32, h=24):
    sd = copy.deepcopy(small_scene(demo2, w, h))
    sd.shapes = list(sd.shapes) + list(meshes)
    return sd


def _one(flux, v0, v1, v2, mat=None):
    from flux_amd.scene import MeshData
    mat = mat or flux.EmissiveData((1.0, 0.5, 0.25), 2.0)
    return MeshData(np.array([v0, v1, v2], dtype=np.float64), np.array([[0, 1, 2]], dtype=np.uint32), mat)


# ------------------------------------------------------------------ oracle KATs (CPU)
def test_triangle_hit_kat(flux, oracle_mod, demo2):
    """Hand-derived: unit right triangle in the z=0 plane, ray along +z from (0.25,0.25,-2): t = 2,
    normal = normalize(e1 x e2) = (0,0,1) (never flipped), two-sided, t > T_MIN, edges inclusive."""
    sd = copy.deepcopy(small_scene(demo2, 8, 6))
    sd.shapes = [_one(flux, (0, 0, 0), (1, 0, 0), (0, 1, 0))]
    o = oracle_mod.Oracle(sd, flux.JobConfiguration(1, 5, 50))
    idx, t, n, p = o.scene_hit((0.25, 0.25, -2.0), (0, 0, 1))
    assert idx == 0 and t == 2.0 and np.array_equal(n, [0, 0, 1]) and np.array_equal(p, [0.25, 0.25, 0.0])
    idx, t, n, _ = o.scene_hit((0.25, 0.25, 2.0), (0, 0, -1))          # from the back: same stored normal
    assert idx == 0 and t == 2.0 and np.array_equal(n, [0, 0, 1])
    assert o.scene_hit((0.75, 0.75, -2.0), (0, 0, 1))[0] == -1           # u+v > 1
    assert o.scene_hit((-0.1, 0.5, -2.0), (0, 0, 1))[0] == -1            # u < 0
    assert o.scene_hit((0.5, 0.0, -2.0), (0, 0, 1))[0] == 0              # on an edge: inclusive
    assert o.scene_hit((0.25, 0.25, -2.0), (1, 0, 0))[0] == -1           # parallel: det == 0
    assert o.scene_hit((0.25, 0.25, 1.0), (0, 0, 1))[0] == -1            # behind
    assert o.scene_hit((0.25, 0.25, -0.0004), (0, 0, 1))[0] == -1        # t <= T_MIN
    # emissive triangle emits towards +z side only (materials.rs:41-50 with the stored normal)
    assert np.array_equal(o.shade((0.25, 0.25, 2.0), (0, 0, -1), 1, 0, 0), [2.0, 1.0, 0.5])
    assert np.array_equal(o.shade((0.25, 0.25, -2.0), (0, 0, 1), 1, 0, 0), [0, 0, 0])


def test_triangle_order_and_ties(flux, oracle_mod, demo2):
    """Shapes precede triangles in hit order; coincident triangles resolve to the lower index."""
    sd = copy.deepcopy(small_scene(demo2, 8, 6))
    a = _one(flux, (0, 0, 0), (1, 0, 0), (0, 1, 0), flux.EmissiveData((1, 0, 0), 1.0))
    b = _one(flux, (0, 0, 0), (1, 0, 0), (0, 1, 0), flux.EmissiveData((0, 1, 0), 1.0))
    pl = flux.PlaneData((0, 0, 0), (0, 0, 1), flux.EmissiveData((0, 0, 1), 1.0))
    sd.shapes = [a, b]
    o = oracle_mod.Oracle(sd, flux.JobConfiguration(1, 5, 50))
    assert o.scene_hit((0.25, 0.25, 2.0), (0, 0, -1))[0] == 0
    sd.shapes = [a, pl, b]  # the plane is an analytic shape: index 0, triangles follow as 1, 2
    o = oracle_mod.Oracle(sd, flux.JobConfiguration(1, 5, 50))
    assert o.scene_hit((0.25, 0.25, 2.0), (0, 0, -1))[0] == 0
    assert np.array_equal(o.shade((0.25, 0.25, 2.0), (0, 0, -1), 1, 0, 0), [0, 0, 1])


def test_yaml_mesh_and_triangle(flux):
    import yaml
    doc = yaml.safe_load(open(__import__("os").path.join(__import__("conftest").SCENES, "demo1.yml")))
    doc["shapes"].append({"Triangle": {"v0": [0, 0, 0], "v1": [1, 0, 0], "v2": [0, 1, 0],
                                       "material": {"Emissive": {"color": [1, 1, 1], "power": 1.0}}}})
    doc["shapes"].append({"Mesh": {"vertices": [[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]],
                                   "triangles": [[0, 1, 2], [1, 3, 2]],
                                   "material": {"Matte": {"diffuse_color": [1, 1, 1], "ambient_color": [1, 1, 1],
                                                          "diffuse_coefficient": 1.0}}}})
    sd = flux.scene_from_dict(doc)
    from flux_amd.scene import MeshData, SceneDesc
    assert isinstance(sd.shapes[-1], MeshData) and sd.shapes[-1].triangles.shape == (2, 3)
    d = SceneDesc(sd)
    assert d.desc.num_shapes == 6 and d.desc.num_meshes == 2 and d.meshes[1].num_triangles == 2
    doc["shapes"][-1]["Mesh"]["triangles"] = [[0, 1, 9]]
    with pytest.raises(flux.SceneError):
        flux.scene_from_dict(doc)


def test_heightfield_generator(flux):
    from flux_amd.procedural import heightfield_mesh, heightfield_scene
    m = heightfield_mesh(10, 5, seed=1)
    assert m.vertices.shape == (11 * 6, 3) and m.triangles.shape == (100, 3)
    assert np.array_equal(m.triangles[0], [0, 1, 7]) and np.array_equal(m.triangles[1], [0, 7, 6])
    assert abs(m.vertices[:, 1]).max() <= 0.37
    m2 = heightfield_mesh(10, 5, seed=1)
    assert np.array_equal(m.vertices, m2.vertices)
    sd = heightfield_scene(4, 4)
    assert len(sd.shapes) == 14 and sd.shapes[-1].triangles.shape == (32, 3)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("nx,nz", [(1, 1), (3, 2), (24, 16), (60, 40)])
@pytest.mark.parametrize("variant", [1, 2])
def test_bvh_equals_bruteforce_equals_oracle(flux, oracle_mod, demo2, nx, nz, variant):
    from flux_amd.procedural import heightfield_scene
    sd = heightfield_scene(nx, nz, seed=7, base=small_scene(demo2, 48, 36))
    cfg = flux.JobConfiguration(8, 5, 50)
    with flux.Renderer(sd, cfg, seed=3) as r:
        r.set_kernel(variant)
        info = r.bvh_info()
        assert info["triangles"] == 2 * nx * nz and info["max_leaf"] <= 4 and info["max_depth"] <= 64
        r.enable_stats(True)
        r.stats(reset=True)
        bvh = r.render_frame()
        st_bvh = r.stats(reset=True)
        r.set_traversal(flux._lib.TRAVERSE_BRUTE)
        brute = r.render_frame()
        st_brute = r.stats(reset=True)
        # the BVH returns exactly the brute-force hits: every path takes the same decisions ...
        for k in ("samples", "segments", "matte_bounces", "glossy_bounces", "specular_bounces", "emissive_hits",
                  "misses", "depth_exhausted"):
            assert st_bvh[k] == st_brute[k], k
        # ... and the same per-sample values; the refill variant's BVH kernel (a traversal state machine,
        # render_body.inc) hands a pixel's samples to lanes in a different order, so its per-pixel SUM may
        # differ in the last bits; the static variant uses one code path for both and is bit-equal
        if variant == 1:
            assert np.array_equal(bvh, brute)
        else:
            assert max_abs_diff(bvh, brute) < 1e-13
        r.set_math(flux.MATH_STRICT)               # reference operation order: one kernel for both, bit-equal
        r.enable_stats(False)
        brute_s = r.render_frame()
        r.set_traversal(flux._lib.TRAVERSE_BVH)
        assert np.array_equal(r.render_frame(), brute_s)
        assert max_abs_diff(brute_s, brute) < 1e-9
    if nx * nz <= 24 * 16:
        want = oracle_mod.Oracle(sd, cfg, seed=3).render_frame(threads=8)
        assert max_abs_diff(bvh, want) < 1e-4
        assert np.percentile(np.abs(bvh - want), 99.9) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("offset", [(5.0e3, 0.0, 0.0), (0.0, -2.0e4, 7.0e3), (1.0e6, 1.0e6, -1.0e6)])
def test_bvh_of_a_mesh_far_from_the_origin(flux, demo2, offset):
    """quantize_bvh's 16-bit grid when the mesh sits hundreds of extents from the origin (ulp_f32(lo) > grid step: qmin
    rounds down by many quanta): every quantised box must still contain its f32 box, i.e. the state-machine
    kernel (DevNodeQ), the inline walk (DevNode) and brute force must agree -- first hits on rays aimed at the mesh, path
    statistics and images of a frame whose camera travels with the mesh."""
    from flux_amd.procedural import heightfield_mesh
    from flux_amd.scene import SphereData, PlaneData
    off = np.array(offset)
    sd = copy.deepcopy(small_scene(demo2, 48, 36))
    mesh = heightfield_mesh(40, 30, seed=3)
    mesh.vertices = mesh.vertices * np.array([0.05, 1.0, 0.05]) + off   # ~1.4 x 0.7 x 1.5 in extent, far away
    moved = []
    for s in sd.shapes:
        s = copy.deepcopy(s)
        if isinstance(s, SphereData):
            s.center = tuple(np.array(s.center) * (0.05 if s.radius < 50 else 1.0) + off)
            s.radius = s.radius * 0.05 if s.radius < 50 else s.radius
        elif isinstance(s, PlaneData):
            s.point = tuple(np.array(s.point) + off - np.array([0.0, 1.0, 0.0]))
        moved.append(s)
    sd.shapes = moved + [mesh]
    sd.camera_settings.eye = tuple(np.array([0.0, 1.6, -1.2]) + off)
    sd.camera_settings.look_at = tuple(np.array([0.0, 0.0, 0.2]) + off)
    rng = np.random.default_rng(9)
    n = 4096
    tgt = mesh.vertices[rng.integers(0, len(mesh.vertices), n)]
    o = tgt + rng.normal(size=(n, 3)) * np.array([1.0, 0.3, 1.0]) + np.array([0.0, 0.8, 0.0])
    d = tgt - o + rng.normal(size=(n, 3)) * 0.02
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    with flux.Renderer(sd, flux.JobConfiguration(8, 5, 50), seed=2) as r:
        r.set_traversal(flux._lib.TRAVERSE_BVH)
        _, hit_b, t_b = r.debug_shade(o, d, depth=5)
        r.set_traversal(flux._lib.TRAVERSE_BRUTE)
        _, hit_f, t_f = r.debug_shade(o, d, depth=5)
        assert np.array_equal(hit_b, hit_f)
        assert (hit_f >= len(moved)).sum() > n // 3                     # a good share of first hits ARE triangles
        assert np.abs(t_b - t_f).max() <= 1e-12 * max(1.0, np.abs(t_f).max())
        r.enable_stats(True)
        out = {}
        for trav in (flux._lib.TRAVERSE_BVH, flux._lib.TRAVERSE_BRUTE):
            r.set_traversal(trav)
            r.stats(reset=True)
            img = r.render_frame()                                      # 64 spp: the state machine over DevNodeQ
            st = r.stats(reset=True)
            out[trav] = (img, {k: v for k, v in st.items() if k not in ("bvh_nodes", "tris_tested")})
        assert out[flux._lib.TRAVERSE_BVH][1] == out[flux._lib.TRAVERSE_BRUTE][1]
        assert max_abs_diff(out[flux._lib.TRAVERSE_BVH][0], out[flux._lib.TRAVERSE_BRUTE][0]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("nx,nz", [(0, 0), (3, 2), (24, 16)])
@pytest.mark.parametrize("where", ["tiny_seen_from_afar", "tiny_at_large_coordinates"])
def test_rays_whose_padding_exceeds_the_grid(flux, oracle_mod, demo2, nx, nz, where):
    """the conservative slab padding of a ray, 2^-20 (bvh_mag + |o|), can exceed the WHOLE 16-bit
    grid of a mesh -- a mesh of 1e-4 seen from 300 units away, or the same mesh at coordinates 1e3 (1e7 times its size) --
    and then every box passes the f32 test, the inverted boxes of a node's EMPTY slots included; their links used to be
    stacked like any other, beyond the per-lane stack (sized for the occupied slots: ONE entry for a one-triangle mesh).
    render_bvh4_kernel now decides per ray whether an empty slot can pass and, if so, walks the tree with constants under
    which exactly the occupied boxes pass (trav_guard4).  The walk, brute force and the oracle must agree: identical path
    statistics, images to rounding -- through kernel 4, with the stack the launch plan really allocates."""
    from flux_amd.procedural import heightfield_mesh
    from flux_amd.scene import MeshData, MatteData
    size = 1.0e-4
    centre = np.array([0.0, 0.0, 0.0]) if where == "tiny_seen_from_afar" else np.array([1.0e3, -2.0e3, 1.5e3])
    if nx == 0:   # one triangle: wide_stack 0, a one-entry stack
        v = np.array([[-0.5, -0.4, 0.0], [0.5, -0.3, 0.1], [0.0, 0.5, -0.1]]) * size + centre
        mesh = MeshData(v, np.array([[0, 1, 2]], dtype=np.uint32), MatteData((0.6, 0.5, 0.4), (0, 0, 0), 0.9))
    else:
        mesh = heightfield_mesh(nx, nz, seed=5)
        # the generator's 28 x 0.7 x 30 field, tilted towards the camera and shrunk to `size`
        x, y, z = mesh.vertices[:, 0] / 30.0, mesh.vertices[:, 1] / 30.0, (mesh.vertices[:, 2] - 5.0) / 30.0
        mesh.vertices = np.stack([x, z * 0.8 + y * 4.0, z * -0.6], axis=-1) * size + centre
    sd = copy.deepcopy(small_scene(demo2, 24, 18))
    env = copy.deepcopy(next(s for s in sd.shapes if type(s).__name__ == "SphereData" and s.invert))
    env.center, env.radius = tuple(centre), 1000.0                # the (emissive) environment, around the mesh and the camera
    sd.shapes = [env, mesh]
    dist = 300.0
    sd.camera_settings.eye = tuple(centre + np.array([0.0, 0.0, -dist]))
    sd.camera_settings.look_at = tuple(centre)
    sd.camera_data.lens_radius = 0.0
    sd.camera_data.focal_distance = dist
    # field of view: 2.5 mesh sizes across the 24 pixels at the mesh's distance
    sd.camera_data.zoom_factor = 24 * sd.output_settings.pixel_size * dist / (sd.camera_data.view_plane_distance * 2.5 * size)
    cfg = flux.JobConfiguration(8, 5, 50)
    o = oracle_mod.Oracle(sd, cfg, seed=6)
    o.stats(reset=True)
    want = o.render_frame(threads=8)
    with flux.Renderer(sd, cfg, seed=6) as r:
        plan, info = r.launch_plan(), r.bvh_info()
        assert plan["kernel"] == flux._lib.PLAN_BVH4
        # the stack this test must not overrun: exactly the tree's bound, with the analytic set's records right behind it (round 5:
        # one environment sphere = a 96-B hit record + a 32-B scan record, and three 64-B materials: sphere, mesh, one spare)
        assert plan["lds"] == max(info["wide_stack"], 1) * 256 + (96 + 32 + 3 * 64)
        r.enable_stats(True)
        out = {}
        for name, trav in (("wide", flux._lib.TRAVERSE_BVH), ("brute", flux._lib.TRAVERSE_BRUTE)):
            r.set_traversal(trav)
            r.stats(reset=True)
            img = r.render_frame()
            out[name] = (img, r.stats(reset=True))
    core = {k: {a: b for a, b in v[1].items() if a not in ("bvh_nodes", "tris_tested")} for k, v in out.items()}
    assert core["wide"] == core["brute"] == {k: core["brute"][k] for k in core["brute"]}
    assert {k: core["wide"][k] for k in o.stats()} == o.stats()
    assert core["wide"]["matte_bounces"] > 24 * 18 * 64 // 20      # the mesh IS hit: a good share of the frame
    assert out["wide"][1]["bvh_nodes"] > 0
    assert max_abs_diff(out["wide"][0], out["brute"][0]) < 1e-12
    assert max_abs_diff(out["wide"][0], want) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("nx,nz,n", [(60, 40, 8), (7, 5, 11)])
def test_wide_and_binary_state_machines_agree(flux, demo2, nx, nz, n):
    """The FAST mesh kernel over the 4-wide tree with quad leaf records (render_bvh4_kernel, the default), the same state
    machine over the binary tree (render_bvh_kernel: the fallback, selected here through FLUX_TRAVERSE_BVH_BINARY) and brute
    force take the same decisions -- identical path statistics -- and give the same image up to the order in which a
    pixel's samples are summed; the node / triangle-test counters show that the two trees really are different walks."""
    from flux_amd.procedural import heightfield_scene
    sd = heightfield_scene(nx, nz, seed=11, base=small_scene(demo2, 48, 36))
    with flux.Renderer(sd, flux.JobConfiguration(n, 5, 50), seed=4) as r:
        assert r.bvh_info()["wide_in_use"] == 1 and r.bvh_info()["leaf_records"] >= nx * nz
        r.enable_stats(True)
        out = {}
        for name, trav in (("wide", flux._lib.TRAVERSE_BVH), ("binary", flux._lib.TRAVERSE_BVH_BINARY), ("brute", flux._lib.TRAVERSE_BRUTE)):
            r.set_traversal(trav)
            r.stats(reset=True)
            img = r.render_frame()
            st = r.stats(reset=True)
            out[name] = (img, st)
        r.set_traversal(flux._lib.TRAVERSE_BVH_BINARY)
        assert r.bvh_info()["wide_in_use"] == 0
    drop = ("bvh_nodes", "tris_tested")
    core = {k: {a: b for a, b in v[1].items() if a not in drop} for k, v in out.items()}
    assert core["wide"] == core["binary"] == core["brute"]
    assert max_abs_diff(out["wide"][0], out["brute"][0]) < 1e-13 and max_abs_diff(out["binary"][0], out["brute"][0]) < 1e-13
    assert 0 < out["wide"][1]["bvh_nodes"] < out["binary"][1]["bvh_nodes"]          # half as deep
    assert out["brute"][1]["bvh_nodes"] == 0 and out["brute"][1]["tris_tested"] == out["brute"][1]["segments"] * 2 * nx * nz


_ENTRY_CAMERAS = {
    "pinhole": dict(lens_radius=0.0),
    "wide_lens": dict(lens_radius=0.8, focal_distance=6.0),
    "lens_past_focus": dict(lens_radius=0.4, focal_distance=1.5),                 # the whole mesh lies beyond the focal plane
    "inside_the_box": dict(eye=(0.5, 0.05, -3.0), look_at=(0.0, 0.0, 8.0)),         # eye between the terrain's lowest and highest point
    "from_below": dict(eye=(2.0, -0.9, 1.0), look_at=(0.0, 0.3, 4.0)),
    "far_zoomed": dict(eye=(0.0, 5.5e4, -9.0e4), look_at=(0.0, 1.0, 0.0), zoom_factor=1.5e4, focal_distance=1.0e5),
    "sideways_up": dict(eye=(-20.0, 0.2, 5.0), look_at=(0.0, 0.0, 5.0), up=(0.0, 0.0, 1.0)),
    "misses_the_mesh": dict(eye=(0.0, 5.0, -9.0), look_at=(0.0, 40.0, -9.5)),
}


@pytest.mark.gpu
@pytest.mark.parametrize("camera", sorted(_ENTRY_CAMERAS))
def test_camera_ray_entry_nodes(flux, demo2, camera):
    """render_bvh4_kernel starts a pixel's camera rays below the part of the tree the pixel's ray bundle (lens x pixel
    footprint) cannot reach (pixel_entry_node).  The skipped boxes must hold nothing any ray of the bundle can hit: for
    cameras that stress the bundle's bounds -- no lens, a wide one, a focal plane in front of the mesh, an eye inside the
    mesh's bounding box, below it, very far away, a rolled view, a view that misses the mesh -- the 4-wide walk, the binary
    walk (which has no entry nodes) and brute force take identical decisions."""
    from flux_amd.procedural import heightfield_scene
    sd = heightfield_scene(60, 40, seed=21, base=small_scene(demo2, 40, 30))
    kw = _ENTRY_CAMERAS[camera]
    cs, cd = sd.camera_settings, sd.camera_data
    cs.eye, cs.look_at, cs.up = kw.get("eye", cs.eye), kw.get("look_at", cs.look_at), kw.get("up", cs.up)
    cd.lens_radius = kw.get("lens_radius", cd.lens_radius)
    cd.focal_distance = kw.get("focal_distance", cd.focal_distance)
    cd.zoom_factor = kw.get("zoom_factor", cd.zoom_factor)
    with flux.Renderer(sd, flux.JobConfiguration(8, 5, 50), seed=9) as r:
        assert r.bvh_info()["wide_in_use"] == 1
        r.enable_stats(True)
        out = {}
        for name, trav in (("wide", flux._lib.TRAVERSE_BVH), ("binary", flux._lib.TRAVERSE_BVH_BINARY), ("brute", flux._lib.TRAVERSE_BRUTE)):
            r.set_traversal(trav)
            r.stats(reset=True)
            img = r.render_frame()
            out[name] = (img, r.stats(reset=True))
    drop = ("bvh_nodes", "tris_tested")
    core = {k: {a: b for a, b in v[1].items() if a not in drop} for k, v in out.items()}
    assert core["wide"] == core["binary"] == core["brute"], camera
    assert max_abs_diff(out["wide"][0], out["brute"][0]) < 1e-13 and max_abs_diff(out["binary"][0], out["brute"][0]) < 1e-13
    if camera == "misses_the_mesh":   # every camera ray's walk ends before its first node
        assert out["wide"][1]["bvh_nodes"] < out["binary"][1]["bvh_nodes"] // 4


@pytest.mark.gpu
def test_bvh_stats_and_degenerate_meshes(flux, oracle_mod, demo2):
    from flux_amd.procedural import heightfield_scene
    from flux_amd.scene import MeshData
    sd = heightfield_scene(40, 30, seed=2, base=small_scene(demo2, 32, 24))
    with flux.Renderer(sd, flux.JobConfiguration(8, 5, 50), seed=1) as r:
        r.enable_stats(True)
        r.stats(reset=True)
        a = r.render_frame()
        st = r.stats(reset=True)
        assert st["bvh_nodes"] > 0 and 0 < st["tris_tested"] < st["segments"] * 2400 // 20
        r.set_traversal(flux._lib.TRAVERSE_BRUTE)
        b = r.render_frame()
        st2 = r.stats()
        assert st2["tris_tested"] == st2["segments"] * 2400 and st2["bvh_nodes"] == 0
        for k in ("samples", "segments", "matte_bounces", "glossy_bounces", "emissive_hits"):
            assert st[k] == st2[k]
        assert max_abs_diff(a, b) < 1e-13  # same hits; the BVH kernel sums a pixel's samples in another order
    # many coincident + zero-area triangles: ties resolve to the lowest index, degenerate ones never hit
    v = np.array([[0, 0.5, 0], [2, 0.5, 0], [0, 0.5, 2], [1, 0.5, 1]], dtype=np.float64)
    t = np.array([[0, 1, 2]] * 40 + [[0, 3, 3]] * 10 + [[0, 2, 1]] * 7, dtype=np.uint32)
    sd2 = copy.deepcopy(small_scene(demo2, 32, 24))
    sd2.shapes = list(sd2.shapes) + [MeshData(v, t, flux.EmissiveData((0.2, 1.0, 0.1), 3.0))]
    cfg = flux.JobConfiguration(4, 5, 50)
    with flux.Renderer(sd2, cfg, seed=1) as r:
        a = r.render_frame()  # 16 spp: static kernel, one code path for both traversals
        r.set_traversal(flux._lib.TRAVERSE_BRUTE)
        assert np.array_equal(a, r.render_frame())
    assert max_abs_diff(a, oracle_mod.Oracle(sd2, cfg, seed=1).render_frame(threads=4)) < 1e-4
    cfg8 = flux.JobConfiguration(8, 5, 50)  # 64 spp: the BVH state-machine kernel on the same degenerate mesh
    with flux.Renderer(sd2, cfg8, seed=1) as r:
        a8 = r.render_frame()
        r.set_traversal(flux._lib.TRAVERSE_BRUTE)
        assert max_abs_diff(a8, r.render_frame()) < 1e-13
    assert max_abs_diff(a8, oracle_mod.Oracle(sd2, cfg8, seed=1).render_frame(threads=4)) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("n,waves", [(64, 1), (96, 2), (128, 4)])
def test_high_spp_kernel_configurations(flux, oracle_mod, demo2, n, waves):
    """Sample counts at which the refill / split kernels run 1, 2 and 4 waves per pixel (K = the largest power of two with
    K * FLUX_MIN_SAMPLES_PER_WAVE = K * 4096 <= N: 4096, 9216 and 16384 spp; the launch planner is ASKED, so this test
    fails if the threshold moves and the counts no longer reach K > 1) while the BVH kernel stays at one: every
    combination launches and agrees (BVH == brute force; FAST == STRICT; vs oracle)."""
    from flux_amd.procedural import heightfield_scene
    sd = heightfield_scene(6, 4, seed=11, base=small_scene(demo2, 12, 8))
    cfg = flux.JobConfiguration(n, 4, 50)
    with flux.Renderer(sd, cfg, seed=2) as r:
        out = {}
        for math in (flux.MATH_FAST, flux.MATH_STRICT):
            for trav in (flux._lib.TRAVERSE_BVH, flux._lib.TRAVERSE_BRUTE):
                r.set_math(math)
                r.set_traversal(trav)
                plan = r.launch_plan()
                if math == flux.MATH_FAST and trav == flux._lib.TRAVERSE_BVH:
                    assert plan["kernel"] == flux._lib.PLAN_BVH4 and plan["waves_per_pixel"] == 1
                else:  # the refill kernel, with the BVH walk or the brute-force scan inside
                    assert plan["kernel"] == flux._lib.PLAN_REFILL and plan["waves_per_pixel"] == waves
                out[(math, trav)] = r.render_frame()
        ref = out[(flux.MATH_STRICT, flux._lib.TRAVERSE_BRUTE)]
        for k, v in out.items():
            assert np.isfinite(v).all() and max_abs_diff(v, ref) < 1e-9, k
    want = oracle_mod.Oracle(sd, cfg, seed=2).render_frame(threads=8)
    assert max_abs_diff(ref, want) < 1e-4
    plain = small_scene(demo2, 12, 8)  # analytic scene at the same sample counts (1/2/4 waves per pixel)
    with flux.Renderer(plain, cfg, seed=2) as r:
        assert r.launch_plan()["kernel"] == flux._lib.PLAN_SPLIT and r.launch_plan()["waves_per_pixel"] == waves
        a = r.render_frame()
        r.set_kernel(flux.KERNEL_REFILL)
        assert r.launch_plan()["kernel"] == flux._lib.PLAN_REFILL and r.launch_plan()["waves_per_pixel"] == waves
        assert max_abs_diff(a, r.render_frame()) < 1e-12
        r.set_kernel(flux.KERNEL_STATIC)
        assert r.launch_plan()["waves_per_pixel"] == 1
        assert max_abs_diff(a, r.render_frame()) < 1e-12
    d = np.abs(a - oracle_mod.Oracle(plain, cfg, seed=2).render_frame(threads=8))
    assert d.max() < 1e-4 and np.percentile(d, 99.9) < 1e-9
