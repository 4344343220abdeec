"""usage: debug_fuzz_case.py <chunk> <case> [old]: re-run one scene of tests/test_gpu_fuzz.py and print oracle vs GPU statistics
for both arithmetics and all kernels (FLUX_HIP_LIB selects an experiment build).  `old`: the scene as the generator drew it
before round 4 (plane normals never normalised; the (chunk, case) pairs of round 3's soaks name those)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flux_amd as flux
from oracle import oracle
from test_gpu_fuzz import random_scene
chunk, want_case = int(sys.argv[1]), int(sys.argv[2])
demo1 = flux.load_scene(os.path.join(ROOT, "scenes", "demo1.yml"))
rng = np.random.default_rng(1000 + chunk)
for case in range(40):
    sd = random_scene(flux, demo1, rng, unit_planes=(case % 2 == 1) and "old" not in sys.argv[3:])
    n = int(rng.choice([1, 2, 3, 8, 9])); D = int(rng.choice([1, 3, 5, 9]))
    seed = int(rng.integers(1, 1 << 30))
    if case != want_case:
        continue
    cfg = flux.JobConfiguration(n, D, 50)
    o = oracle.Oracle(sd, cfg, seed=seed); o.stats(reset=True); want = o.render_frame(threads=4); ost = o.stats()
    print("scene", sd.output_settings.image_width, sd.output_settings.image_height, "n", n, "D", D, "lens", sd.camera_data.lens_radius)
    for s in sd.shapes: print("  ", s)
    print("oracle", ost)
    with flux.Renderer(sd, cfg, seed=seed) as r:
        for math in (flux.MATH_FAST, flux.MATH_STRICT):
            r.set_math(math)
            for variant in (1, 2, 3):
                r.set_kernel(variant); r.enable_stats(True); r.stats(reset=True)
                got = r.render_frame(); st = r.stats()
                d = {k: st[k] - ost[k] for k in ost if st[k] != ost[k]}
                fin = np.isfinite(want) & np.isfinite(got)
                nan_diff = np.argwhere(np.isfinite(want) != np.isfinite(got))
                print("math", math, "variant", variant, "diff", d, "max|d|", float(np.abs(got[fin] - want[fin]).max(initial=0.0)),
                      "non-finite: oracle", int((~np.isfinite(want)).sum()), "gpu", int((~np.isfinite(got)).sum()),
                      "where they differ", [(tuple(int(x) for x in ix), float(want[tuple(ix)]), float(got[tuple(ix)])) for ix in nan_diff[:6]])
