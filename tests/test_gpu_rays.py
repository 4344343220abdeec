"""Ray-level parity: Scene::hit / Scene::shade on the device (flux_debug_shade) against the oracle for
individual rays -- hand-picked ones that exercise the reference's corner cases (SURVEY.md 8c KAT list) and
tens of thousands of random ones -- in both arithmetics.  Decisions (which shape, or none) must be identical;
distances and radiance agree to rounding."""
import copy

import numpy as np
import pytest

from conftest import small_scene

pytestmark = pytest.mark.gpu

MODES = ["fast", "strict"]


def _mode(flux, name):
    return flux.MATH_FAST if name == "fast" else flux.MATH_STRICT


def _oracle_rays(o, origins, dirs, depth, set_index, sample_index):
    hits, ts, rgbs = [], [], []
    for a, b in zip(origins, dirs):
        idx, t, _, _ = o.scene_hit(a, b)
        hits.append(idx)
        ts.append(t if idx >= 0 else 0.0)
        rgbs.append(o.shade(a, b, depth, set_index, sample_index))
    return np.array(hits), np.array(ts), np.array(rgbs)


def _check(flux, oracle_mod, sd, origins, dirs, math, depth=1, D=5, n=4, set_index=1, sample_index=3, rgb_tol=1e-9):
    cfg = flux.JobConfiguration(n, D, 50)
    o = oracle_mod.Oracle(sd, cfg, seed=4)
    want_hit, want_t, want_rgb = _oracle_rays(o, origins, dirs, depth, set_index, sample_index)
    with flux.Renderer(sd, cfg, seed=4) as r:
        r.set_math(_mode(flux, math))
        rgb, hit, t = r.debug_shade(origins, dirs, depth, set_index, sample_index)
    assert np.array_equal(hit, want_hit), np.nonzero(hit != want_hit)[0][:10]
    fin = np.isfinite(want_t)
    assert np.array_equal(np.isfinite(t), fin)
    assert np.allclose(t[fin], want_t[fin], rtol=1e-12, atol=1e-12)
    assert np.array_equal(t[~fin], want_t[~fin])  # +inf where the reference says +inf
    ok = np.isfinite(want_rgb).all(axis=1)
    if math == "strict":
        assert np.array_equal(np.isfinite(rgb).all(axis=1), ok)
    assert np.abs(rgb[ok] - want_rgb[ok]).max(initial=0.0) < rgb_tol
    return hit, t, rgb


@pytest.mark.parametrize("math", MODES)
def test_reference_corner_cases(flux, oracle_mod, demo2, math):
    """Plane::hit's +-inf (shapes.rs:135-152), BoundingBox::hit with zero direction components and origins on a slab
    plane (shapes.rs:98-133), origins on / inside spheres, tangent rays, the inverted sphere, T_MIN."""
    sd = copy.deepcopy(small_scene(demo2, 16, 12))
    s3 = 1.0 / np.sqrt(3.0)
    rays = [
        # parallel to the floor plane (point 0, normal +y): numerator > 0 -> t = +inf, a "hit" at infinity;
        ((0.0, -1.0, 0.0), (1.0, 0.0, 0.0)),
        ((0.0, 1.0, 30.0), (1.0, 0.0, 0.0)),     # numerator < 0 -> t = -inf: no plane hit, the environment sphere wins
        ((0.0, 0.0, 30.0), (0.0, 0.0, 1.0)),     # IN the plane, parallel: 0/0 = NaN: no plane hit
        ((0.0, 5.0, 0.0), (0.0, -1.0, 0.0)),     # straight down onto the sphere at (0,1,0): t = 3
        ((0.0, 2.0, 0.0), (0.0, 1.0, 0.0)),      # from that sphere's pole outward: must not re-hit it (T_MIN)
        ((0.0, 2.0, 0.0), (0.0, -1.0, 0.0)),     # from the pole inward: far root t = 2
        ((0.0, 1.0, 0.0), (s3, s3, s3)),         # from the centre of a unit sphere: exits at t = 1
        ((-5.0, 2.0, 0.0), (1.0, 0.0, 0.0)),     # tangent to the sphere at (0,1,0) (disc == 0 up to rounding)
        ((-5.0, 2.0 + 1e-12, 0.0), (1.0, 0.0, 0.0)),
        ((1.0, 1.0, -20.0), (0.0, 0.0, 1.0)),    # axis-aligned through several spheres' boxes: 1/0 = inf slabs
        ((1.0, 1.0, 2.0 - 1.0), (1.0, 0.0, 0.0)),  # origin exactly on a box face of the sphere at (1,1,2): 0*inf
        ((0.0, 50.0, 0.0), (0.0, 1.0, 0.0)),     # straight up into the inverted environment sphere: t = 50
        ((0.0, 150.0, 0.0), (0.0, 1.0, 0.0)),    # outside the environment sphere looking away: miss -> background
        ((0.0, 150.0, 0.0), (0.0, -1.0, 0.0)),   # outside looking in: hits its outer side (emits only from inside)
        ((-9.0, 7.0, 8.0 - 5.0 - 1e-9), (0.0, 0.0, 1.0)),  # just outside the light sphere: distance below T_MIN
        # BoundingBox::hit's z-slab NaN (shapes.rs:121-130): direction.z == 0 and the origin ON a z face of the box of the
        # sphere at (1,1,2): (corner.z - oz) * (1/0) = 0 * inf = NaN comes out of the reference's max/min forms as t0 resp.
        # t1, `t0 < t1` is false and the sphere is MISSED although the quadratic has disc = 0, t = 6 (the ray is tangent).
        # The same tangent through an x or y face hits: those NaNs are dropped by the max/min forms.
        ((-5.0, 1.0, 1.0), (1.0, 0.0, 0.0)),     # tz_min NaN (corner0.z == oz)
        ((-5.0, 1.0, 3.0), (1.0, 0.0, 0.0)),     # tz_max NaN (corner1.z == oz)
        ((-5.0, 1.0, 3.0), (1.0, 0.0, -0.0)),    # 1/dz = -inf: the slabs swap, still NaN
        ((-5.0, 1.0, 1.0), (-1.0, 0.0, 0.0)),    # looking away: a miss either way
        ((1.0, 0.0 + 1e-3, 2.0 - 6.0), (0.0, 0.0, 1.0)),  # control: tangent at the sphere's lowest point through its y face
        ((1.0, 2.0, 2.0 - 6.0), (0.0, 0.0, 1.0)),          # control: exactly ON the y face (ty NaN is dropped): tangent, decided by the quadratic
        ((2.0, 1.0, 2.0 - 6.0), (0.0, 0.0, 1.0)),          # control: exactly ON the x face
    ]
    origins = np.array([a for a, _ in rays])
    dirs = np.array([b for _, b in rays])
    hit, t, rgb = _check(flux, oracle_mod, sd, origins, dirs, math)
    assert hit[0] == 0 and hit[1] == 0 and hit[2] == 0   # an infinitely distant plane hit loses to the environment sphere
    assert hit[3] == 4 and abs(t[3] - 3.0) < 1e-12
    assert hit[5] == 4 and abs(t[5] - 2.0) < 1e-12 and hit[6] == 4 and abs(t[6] - 1.0) < 1e-12
    assert hit[11] == 0 and abs(t[11] - 50.0) < 1e-12 and hit[12] == -1 and hit[13] == 0
    assert np.array_equal(rgb[12], sd.background)
    assert hit[15] != 5 and hit[16] != 5 and hit[17] != 5  # the z-slab NaN: sphere #5 (centre (1,1,2)) is missed, as in the reference
    # the plane alone: the parallel ray with a positive numerator "hits" it at t = +inf (Plane::hit's quirk)
    only_plane = copy.deepcopy(sd)
    only_plane.shapes = [sd.shapes[12]]
    hit, t, _ = _check(flux, oracle_mod, only_plane, origins[:3], dirs[:3], math)
    assert hit[0] == 0 and np.isposinf(t[0]) and hit[1] == -1 and hit[2] == -1


@pytest.mark.parametrize("math", MODES)
def test_tie_break_and_nested_shapes(flux, oracle_mod, demo1, math):
    sd = copy.deepcopy(small_scene(demo1, 16, 12))
    base = sd.shapes[1]
    twin = copy.deepcopy(base)
    twin.material = flux.EmissiveData((1.0, 0.0, 0.0), 5.0)
    inner = flux.SphereData(base.center, base.radius * 0.5, flux.EmissiveData((0.0, 1.0, 0.0), 2.0), False)
    sd.shapes = [sd.shapes[0], base, twin, inner] + list(sd.shapes[2:])
    c = np.array(base.center)
    origins = np.array([c + (0, 5, 0), c + (0, 0.75, 0), c, c + (3, 0, 0)])
    dirs = np.array([(0, -1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0)], dtype=np.float64)
    hit, t, _ = _check(flux, oracle_mod, sd, origins, dirs, math)
    assert hit[0] == 1            # coincident twin (index 2) never wins
    assert hit[1] == 3 and hit[2] == 3  # from inside the shell: the inner sphere first
    assert hit[3] == 1


@pytest.mark.parametrize("math", MODES)
@pytest.mark.parametrize("depth", [1, 3, 5, 6])
def test_random_rays(flux, oracle_mod, demo2, math, depth):
    """20 000 random rays per (mode, depth) in the demo2 geometry: origins in the room, on the floor and on sphere
    surfaces; unit, axis-aligned and NON-unit directions (a reflection off a non-unit normal produces those)."""
    rng = np.random.default_rng(100 + depth)
    sd = small_scene(demo2, 16, 12)
    n = 20000
    origins = rng.uniform([-8, 0.01, -8], [10, 8, 16], (n, 3))
    origins[: n // 5, 1] = 0.0                                           # on the floor plane
    k = n // 5
    centres = np.array([s.center for s in sd.shapes[2:12]])
    pick = rng.integers(0, len(centres), k)
    u = rng.normal(size=(k, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    origins[k:2 * k] = centres[pick] + u                                 # on a unit sphere's surface (to rounding)
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs[::7] *= rng.uniform(0.2, 3.0, (len(dirs[::7]), 1))              # non-unit
    axis = rng.integers(0, 3, n // 50)
    dirs[: n // 50] = np.eye(3)[axis] * rng.choice([-1.0, 1.0], (n // 50, 1))  # exact zeros in two components
    _check(flux, oracle_mod, sd, origins, dirs, math, depth=depth, rgb_tol=1e-8)


@pytest.mark.parametrize("math", MODES)
def test_random_rays_with_triangles(flux, oracle_mod, demo2, math):
    from flux_amd.procedural import heightfield_scene
    sd = heightfield_scene(12, 8, seed=5, base=small_scene(demo2, 16, 12))
    rng = np.random.default_rng(77)
    n = 6000
    origins = rng.uniform([-8, 0.3, -8], [10, 8, 16], (n, 3))
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    hit, _, _ = _check(flux, oracle_mod, sd, origins, dirs, math, rgb_tol=1e-8)
    assert (hit >= len(sd.shapes) - 1).sum() > 100  # a good share of first hits are triangles


@pytest.mark.parametrize("math", MODES)
def test_grazing_rays_through_the_f32_filter(flux, oracle_mod, demo2, math):
    """FAST's candidate filter runs in packed f32 with a conservative bias: it may let extra spheres
    through but must never drop one the exact f64 test would hit.  Rays aimed past every sphere of demo2 with an impact
    parameter r (1 -/+ eps) -- from near and far origins, towards and away from the sphere, from inside the environment
    sphere's shell -- must hit exactly what the oracle says (first-hit id, distance).  eps runs from 1e-3 down to 1e-15
    in STRICT (the oracle's own formula: decisions agree to the last bit) and down to 1e-11 in FAST, whose half-b
    discriminant along the unit direction rounds differently from b^2 - 4ac once the discriminant 2 r^2 eps is within
    a few ulp of |o - p|^2 (observed: the two disagree only at eps <= 1e-14)."""
    from flux_amd.scene import SphereData
    rng = np.random.default_rng(11)
    origins, dirs = [], []
    for s in demo2.shapes:
        if not isinstance(s, SphereData):
            continue
        c, r = np.array(s.center, dtype=np.float64), float(s.radius)
        for eps in 10.0 ** -np.arange(3, 16 if math == "strict" else 12):
            for sign in (-1.0, 1.0):
                for dist in (1.5, 7.0, 40.0):
                    u = rng.normal(size=3)
                    u /= np.linalg.norm(u)
                    v = np.cross(u, rng.normal(size=3))
                    v /= np.linalg.norm(v)
                    o = c - u * (r + dist) + v * r * (1.0 + sign * eps)   # passes the centre at distance r (1 +/- eps)
                    if s.invert and np.linalg.norm(o) >= 99.0:
                        o = c + v * r * (1.0 - eps)                      # the environment sphere: start just inside its shell
                    origins.append(o)
                    dirs.append(u)
                    origins.append(o)
                    dirs.append(-u)                                      # and the same line the other way round
    origins, dirs = np.array(origins), np.array(dirs)
    sd = copy.deepcopy(small_scene(demo2, 16, 12))
    cfg = flux.JobConfiguration(4, 5, 50)
    o = oracle_mod.Oracle(sd, cfg, seed=4)
    want = [o.scene_hit(a, b) for a, b in zip(origins, dirs)]
    want_hit = np.array([w[0] for w in want])
    want_t = np.array([w[1] if w[0] >= 0 else 0.0 for w in want])
    with flux.Renderer(sd, cfg, seed=4) as r:
        r.set_math(_mode(flux, math))
        _, hit, t = r.debug_shade(origins, dirs, 5, 1, 3)
    assert np.array_equal(hit, want_hit), np.nonzero(hit != want_hit)[0][:10]
    # a grazing hit's distance -hb -/+ sqrt(dq) is ill-conditioned in dq (d t / d dq = 1 / (2 sqrt(dq))): STRICT shares the
    # oracle's rounding exactly, FAST's other formula moves t by up to ~1e-13 / sqrt(2 r^2 eps)
    assert np.abs(t - want_t).max() <= (0.0 if math == "strict" else 1e-6)
    assert len(set(hit.tolist())) >= 10       # the rays really do reach most of the scene's shapes
