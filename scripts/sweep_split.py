"""Time kernel variant 3 (split) and 2 (refill) for every libflux_hip_*.so under flux_amd/variants plus the default build.
usage (GPU box): python scripts/sweep_split.py [root]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1] if len(sys.argv) > 1 else "128"
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "flux_amd", "variants", "*.so")))
for lib in libs:
    if lib and "clock" in lib:
        continue
    print(os.path.basename(lib) if lib else "default", flush=True)
    env = dict(os.environ)
    if lib:
        env["FLUX_HIP_LIB"] = lib
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "quick_time.py"), "demo2", root, "2,3"], env=env,
                       capture_output=True, text=True)
    print("\n".join(l for l in p.stdout.splitlines() if "rep 1" in l), flush=True)
