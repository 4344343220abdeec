"""Time the 1M-triangle configuration for the default build and every variant library.  usage: sweep_hf.py [root]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1] if len(sys.argv) > 1 else "32"
for lib in [None] + sorted(glob.glob(os.path.join(ROOT, "flux_amd", "variants", "*.so"))):
    print(os.path.basename(lib) if lib else "default", flush=True)
    env = dict(os.environ)
    if lib:
        env["FLUX_HIP_LIB"] = lib
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "quick_time.py"), "hf:1000x500", root, "0", "stats"], env=env,
                       capture_output=True, text=True)
    print("\n".join(l[:230] for l in p.stdout.splitlines() if "rep 1" in l or "stats" in l or "bvh" in l), p.stderr[-300:] if p.returncode else "", flush=True)
