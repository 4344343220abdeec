"""Time the 1M-triangle configuration for the default build and every variant library (except t_trips: the lane census).
usage: sweep_hf.py [root]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1] if len(sys.argv) > 1 else "32"
for lib in [None] + sorted(glob.glob(os.path.join(ROOT, "flux_amd", "variants", "*.so"))):
    name = os.path.basename(lib) if lib else "default"
    env = dict(os.environ)
    if lib:
        env["FLUX_HIP_LIB"] = lib
    if "trips" in name:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bvh_lanes.py"), "1000x500", "16"], env=env, capture_output=True, text=True)
        print(name, "\n" + p.stdout, p.stderr[-300:] if p.returncode else "", flush=True)
        continue
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "quick_time.py"), "hf:1000x500", root, "0", "stats"], env=env,
                       capture_output=True, text=True)
    ms = re.findall(r"rep 1: kernel ([0-9.]+) ms", p.stdout)
    st = re.findall(r"per segment: nodes ([0-9.]+) tris ([0-9.]+)", p.stdout)
    mean = re.findall(r"rep 1:.*mean=([0-9.]+)", p.stdout)
    print(f"{name:18s} kernel {ms[0] if ms else '?':>8s} ms   nodes/segment {st[0][0] if st else '?'}  tris/segment {st[0][1] if st else '?'}  mean {mean[0] if mean else '?'}",
          p.stderr[-300:] if p.returncode else "", flush=True)
