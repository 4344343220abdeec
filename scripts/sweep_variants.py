"""Build experiment variants of libflux_hip.so (here, no GPU needed) -> build/variants/*.so.
Then on the GPU box: scripts/sweep_variants.py --run  times each with quick_time.py and checks parity
against the default build's image."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR_DIR = os.path.join(ROOT, "flux_amd", "variants")
VARIANTS = {
    "k4": [],
    "k8": ["-DFLUX_MAX_WAVES_PER_PIXEL=8"],
    "k2": ["-DFLUX_MAX_WAVES_PER_PIXEL=2"],
}
if "--run" not in sys.argv:
    from flux_amd import build
    os.makedirs(VAR_DIR, exist_ok=True)
    names = [a for a in sys.argv[1:] if a in VARIANTS] or list(VARIANTS)
    for name in names:
        out = os.path.join(VAR_DIR, f"libflux_hip_{name}.so")
        build.build_variant(out, VARIANTS[name] + ["-Rpass-analysis=kernel-resource-usage"], verbose=False)
        print("built", out)
else:
    import glob
    code = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
import flux_amd
scene = os.environ.get("SWEEP_SCENE", "demo2")
if scene.startswith("hf:"):
    from flux_amd.procedural import heightfield_scene
    nx, nz = [int(x) for x in scene[3:].split("x")]
    sd = heightfield_scene(nx, nz)
else:
    sd = flux_amd.load_scene(os.path.join(%r, "scenes", scene + ".yml"))
n = int(os.environ.get("SWEEP_ROOT", "32"))
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
for v in (2,):
    r.set_kernel(v)
    best = 1e9
    for _ in range(3):
        img = r.render_frame(); best = min(best, r.last_kernel_ms())
    ref_path = "/tmp/sweep_ref_%%s_%%d.npy" %% (scene.replace(":", "_"), n)
    if not os.path.exists(ref_path): np.save(ref_path, img)
    err = float(np.abs(img - np.load(ref_path)).max())
    print("  variant %%d: %%8.2f ms  %%8.1f Msamples/s  max|d| vs first = %%.3e" %% (v, best, 800*600*n*n/best/1e3, err), flush=True)
''' % (ROOT, ROOT)
    for lib in sorted(glob.glob(os.path.join(VAR_DIR, "*.so")), key=lambda p: (not p.endswith("_base.so"), p)):
        print(os.path.basename(lib), flush=True)
        env = dict(os.environ, FLUX_HIP_LIB=lib)
        subprocess.run([sys.executable, "-c", code], env=env, check=False)
