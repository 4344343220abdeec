"""Differential fuzzing: random scenes (cameras with and without depth of field, nested / inverted / tiny / huge
spheres, tilted planes, every material with awkward parameters, coloured backgrounds) rendered by the
HIP path in both arithmetics and all kernels against the oracle.  Besides the image tolerance, path
statistics must match exactly -- any ray/primitive or material decision that differs shows up there.

Plane normals: the reference never normalises them (shapes.rs:135-152).  Every EVEN case keeps the generator's non-unit
normals: reflected directions are then non-unit, Phong lobes under- and overflow, and the reference's recursion produces
NaN pixels; FLUX_MATH_FAST is not defined there and the library renders such a scene with the STRICT arithmetic whatever
the setting (abi.hip effective_math; the launch plan says so).  Every ODD case has its plane normals normalised: the
scenes FLUX_MATH_FAST is defined for, with all of its shortcuts active (unit directions, self-skip, closed-form weights) --
until round 4 the generator made practically every scene of the first kind, so those shortcuts were barely fuzzed."""
import os

import numpy as np
import pytest

import copy

from conftest import max_abs_diff, small_scene

pytestmark = pytest.mark.gpu


def random_scene(flux, base, rng, unit_planes=False):
    """`unit_planes` normalises the plane normals AFTER the scene is drawn, so the random stream -- and with it every scene a
    (chunk, case) pair of an earlier soak names -- is the same either way."""
    sd = copy.deepcopy(base)
    W, H = int(rng.integers(8, 40)), int(rng.integers(6, 30))
    sd.output_settings.image_width, sd.output_settings.image_height = W, H
    sd.output_settings.pixel_size = float(rng.uniform(4.0, 20.0))
    sd.background = tuple(float(x) for x in rng.uniform(0.0, 1.5, 3)) if rng.uniform() < 0.5 else (0.0, 0.0, 0.0)
    eye = rng.uniform(-6, 6, 3)
    eye[1] = abs(eye[1]) + 0.5
    sd.camera_settings.eye = tuple(float(x) for x in eye)
    sd.camera_settings.look_at = tuple(float(x) for x in rng.uniform(-1, 1, 3))
    sd.camera_settings.up = (0.0, 1.0, 0.0) if rng.uniform() < 0.7 else tuple(float(x) for x in rng.normal(size=3))
    sd.camera_data.lens_radius = float(rng.choice([0.0, 0.05, 0.3]))
    sd.camera_data.focal_distance = float(rng.uniform(3.0, 12.0))
    sd.camera_data.zoom_factor = float(rng.uniform(0.5, 2.0))

    def color():
        return tuple(float(x) for x in rng.uniform(0.0, 1.0, 3))

    def material():
        k = int(rng.integers(0, 4))
        if k == 0:
            return flux.MatteData(color(), color(), float(rng.uniform(0.1, 1.0)))
        if k == 1:
            return flux.EmissiveData(color(), float(rng.uniform(0.0, 4.0)))
        if k == 2:
            return flux.ReflectiveData(float(rng.uniform(0.1, 1.0)), color())
        return flux.GlossyReflectiveData(float(rng.uniform(0.1, 1.0)), color(),
                                         float(rng.choice([0.0, 1.0, 2.5, 10.0, 100.0, 1e4, 1e5])))

    shapes = []
    if rng.uniform() < 0.7:
        shapes.append(flux.SphereData((0.0, 0.0, 0.0), float(rng.uniform(20, 200)), flux.EmissiveData(color(), 1.0), True))
    for _ in range(int(rng.integers(0, 40))):
        kind = rng.uniform()
        if kind < 0.8:
            r = float(rng.choice([rng.uniform(0.05, 0.3), rng.uniform(0.3, 2.0), rng.uniform(2.0, 8.0)]))
            shapes.append(flux.SphereData(tuple(float(x) for x in rng.uniform(-5, 5, 3)), r, material(), bool(rng.uniform() < 0.15)))
            if rng.uniform() < 0.1:
                shapes.append(copy.deepcopy(shapes[-1]))  # coincident twin
                shapes[-1].material = material()
        else:
            n = rng.normal(size=3) * rng.choice([0.3, 1.0, 2.5])  # never normalised by the reference
            shapes.append(flux.PlaneData(tuple(float(x) for x in rng.uniform(-3, 3, 3)), tuple(float(x) for x in n), material()))
    if unit_planes:
        for s in shapes:
            if isinstance(s, flux.PlaneData):
                n = np.array(s.normal, dtype=np.float64)
                s.normal = tuple(float(x) for x in n / np.linalg.norm(n))
    sd.shapes = shapes
    return sd


def has_non_unit_plane(flux, sd):
    """abi.hip's rule (DevHitRec::unit_normal): |n.n - 1| <= 4 eps."""
    return any(isinstance(s, flux.PlaneData) and abs(float(np.dot(s.normal, s.normal)) - 1.0) > 4.0 * 2.220446049250313e-16
               for s in sd.shapes)


def check_scene(flux, oracle_mod, sd, n, D, seed, tag0):
    """One scene, both arithmetics, three kernels, against the oracle: statistics equal, NaN pixels the reference's, finite
    pixels within the north-star tolerance."""
    cfg = flux.JobConfiguration(n, D, 50)
    o = oracle_mod.Oracle(sd, cfg, seed=seed)
    o.stats(reset=True)
    want = o.render_frame(threads=4)
    ost = o.stats()
    o.close()
    finite = np.isfinite(want)
    non_unit = has_non_unit_plane(flux, sd)
    with flux.Renderer(sd, cfg, seed=seed) as r:
        for math in (flux.MATH_FAST, flux.MATH_STRICT):
            if D > 31 and math == flux.MATH_STRICT:   # STRICT's recursion stack: 32 B of LDS per level and lane, 31 levels
                continue
            r.set_math(math)
            # which arithmetic really runs: FAST only where it is defined (module docstring) -- unless the job is too deep for
            # STRICT to run at all: then FAST with the long-form glossy weights (abi.hip effective_math)
            assert r.launch_plan()["math"] == (flux.MATH_STRICT if (non_unit and D <= 31) else math), tag0
            for variant in (flux.KERNEL_STATIC, flux.KERNEL_REFILL, flux.KERNEL_SPLIT):
                r.set_kernel(variant)
                r.enable_stats(True)
                r.stats(reset=True)
                got = r.render_frame()
                st = r.stats()
                tag = f"{tag0} math {math} variant {variant} n {n} D {D} shapes {len(sd.shapes)}"
                assert {k: st[k] for k in ost} == ost, tag
                # Where the reference's own arithmetic breaks down -- a Phong lobe (r.wi)^e that under- or overflows, which
                # needs a non-unit plane normal (for unit normals lobe >= 1 - y) -- its long form f (n.wi)/pdf yields NaN
                # (0 * inf) or inf; those pixels must be NaN / inf here too, in both settings
                assert np.array_equal(np.isfinite(got), finite), tag
                assert non_unit or finite.all(), tag      # a scene of unit normals has no such pixel
                if finite.any():
                    assert max_abs_diff(got[finite], want[finite]) < 1e-4, tag


@pytest.mark.parametrize("chunk", range(int(os.environ.get("FLUX_FUZZ_CHUNKS", "8"))))   # a longer soak: FLUX_FUZZ_CHUNKS=80
def test_random_scenes_against_the_oracle(flux, oracle_mod, demo1, chunk):
    rng = np.random.default_rng(1000 + chunk)
    # FLUX_FUZZ_COLLECT=1 (the long soaks): a scene that differs does not end its chunk -- the remaining scenes are still
    # checked and the chunk fails at its END with every differing scene listed, so a soak's count of differing scenes is exact
    # (until round 4 a chunk stopped at its first difference and the count was a lower bound)
    collect = os.environ.get("FLUX_FUZZ_COLLECT") == "1"
    differing = []
    for case in range(40):
        sd = random_scene(flux, demo1, rng, unit_planes=case % 2 == 1)
        n = int(rng.choice([1, 2, 3, 8, 9]))
        D = int(rng.choice([1, 3, 5, 9]))
        seed = int(rng.integers(1, 1 << 30))
        if not collect:
            check_scene(flux, oracle_mod, sd, n, D, seed, f"chunk {chunk} case {case}")
            continue
        try:
            check_scene(flux, oracle_mod, sd, n, D, seed, f"chunk {chunk} case {case}")
        except AssertionError as e:
            differing.append(str(e).splitlines()[0][:200] + " || " + " ".join(l.strip() for l in str(e).splitlines()[1:8] if "!=" in l)[:300])
    assert not differing, f"{len(differing)} DIFFERING SCENE(S) in chunk {chunk}: " + " ## ".join(differing)


# The nine scenes in which FLUX_MATH_FAST differed from the reference in round 3's 240 000-scene soak of the old generator
# (FLUX_FUZZ_CHUNKS=6000; profiles/r03_experiments/fuzz_soak_r03e_summary.log), as it drew them (non-unit plane normals): in
# 4301/23, 4508/18 and 5245/20 one channel of one pixel was finite where the reference holds NaN (FAST multiplied a path's
# throughput front to back, so an overflowing product met its zero in another order than the reference's recursion,
# materials.rs:31-33, 69-71); in the other six, one to three grazing mirror segments were decided the other way by the
# half-b discriminant's rounding.  All nine have a plane with a non-unit normal, so they are rendered with the STRICT
# arithmetic now and must equal the oracle exactly.
SOAK_R03E = [(154, 38), (1399, 31), (2310, 6), (2632, 11), (2795, 15), (4208, 6), (4301, 23), (4508, 18), (5245, 20)]


@pytest.mark.parametrize("chunk,want_case", SOAK_R03E)
def test_the_soak_scenes_in_which_fast_differed(flux, oracle_mod, demo1, chunk, want_case):
    rng = np.random.default_rng(1000 + chunk)
    for case in range(want_case + 1):
        sd = random_scene(flux, demo1, rng)
        n = int(rng.choice([1, 2, 3, 8, 9]))
        D = int(rng.choice([1, 3, 5, 9]))
        seed = int(rng.integers(1, 1 << 30))
    assert has_non_unit_plane(flux, sd)
    check_scene(flux, oracle_mod, sd, n, D, seed, f"soak chunk {chunk} case {want_case}")


@pytest.mark.parametrize("chunk", range(int(os.environ.get("FLUX_FUZZ_MESH_CHUNKS", "4"))))
def test_random_meshes_against_the_oracle(flux, oracle_mod, demo1, chunk):
    """The same with triangle soups added (extension): the BVH state-machine kernel (64 spp), the inline BVH
    (static kernel, STRICT) and brute force must all take the oracle's decisions."""
    from flux_amd.scene import MeshData
    rng = np.random.default_rng(5000 + chunk)
    for case in range(10):
        sd = random_scene(flux, demo1, rng, unit_planes=case % 2 == 1)
        shapes = list(sd.shapes)
        for _ in range(int(rng.integers(1, 4))):
            nv = int(rng.integers(3, 60))
            v = rng.uniform(-4, 4, (nv, 3))
            v[:, 1] = np.abs(v[:, 1]) * 0.5
            nt = int(rng.integers(1, 150))
            # distinct triangles only: two coincident triangles with permuted vertex order have t equal up to
            # rounding and possibly opposite normals, so which one wins (and where a Matte bounce goes) is decided
            # by the last bit -- ill-posed in the oracle itself.  Repeated-vertex (degenerate) triangles are kept:
            # they have no surface and must never be hit.
            seen, tl = set(), []
            for tri in rng.integers(0, nv, (nt, 3)):
                key = frozenset(int(x) for x in tri)
                if len(key) == 3 and key in seen:
                    continue
                seen.add(key)
                tl.append(tri)
            t = np.array(tl, dtype=np.uint32).reshape(-1, 3)
            k = int(rng.integers(0, 3))
            mat = (flux.MatteData((0.6, 0.5, 0.4), (0, 0, 0), 0.9) if k == 0 else
                   flux.EmissiveData((0.3, 0.9, 0.4), 2.0) if k == 1 else
                   flux.GlossyReflectiveData(0.7, (0.9, 0.9, 1.0), 50.0))
            shapes.append(MeshData(v, t, mat))
        sd.shapes = shapes
        n = int(rng.choice([2, 8]))
        cfg = flux.JobConfiguration(n, int(rng.choice([2, 5])), 50)
        seed = int(rng.integers(1, 1 << 30))
        o = oracle_mod.Oracle(sd, cfg, seed=seed)
        o.stats(reset=True)
        want = o.render_frame(threads=4)
        ost = o.stats()
        finite = np.isfinite(want)
        with flux.Renderer(sd, cfg, seed=seed) as r:
            for math in (flux.MATH_FAST, flux.MATH_STRICT):
                r.set_math(math)
                for variant in (flux.KERNEL_STATIC, flux.KERNEL_REFILL):
                    for trav in (flux._lib.TRAVERSE_BVH, flux._lib.TRAVERSE_BRUTE):
                        r.set_kernel(variant)
                        r.set_traversal(trav)
                        r.enable_stats(True)
                        r.stats(reset=True)
                        got = r.render_frame()
                        st = r.stats()
                        tag = f"chunk {chunk} case {case} math {math} variant {variant} trav {trav} n {n}"
                        assert {k: st[k] for k in ost} == ost, tag
                        assert np.array_equal(np.isfinite(got), finite), tag   # NaN pixels: the reference's (check_scene)
                        if finite.any():
                            assert max_abs_diff(got[finite], want[finite]) < 1e-4, tag
