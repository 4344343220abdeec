"""Quick kernel timing on the GPU box (development aid, not the bench contract).
usage: quick_time.py <demo1|demo2|hf:NXxNZ> <roots,comma> <variants,comma> [stats]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flux_amd

scene = sys.argv[1] if len(sys.argv) > 1 else "demo2"
roots = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["8", "32"])]
variants = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "2"])]
want_stats = len(sys.argv) > 4
if scene.startswith("hf:"):
    from flux_amd.procedural import heightfield_scene
    nx, nz = [int(x) for x in scene[3:].split("x")]
    t = time.time(); sd = heightfield_scene(nx, nz); print(f"generated {2*nx*nz} triangles in {time.time()-t:.2f} s")
else:
    sd = flux_amd.load_scene(f"scenes/{scene}.yml")
W, H = sd.output_settings.image_width, sd.output_settings.image_height
for n in roots:
    t = time.time()
    r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
    t_create = time.time() - t
    print(f"{scene} n={n} spp={n*n}: ctx_create {t_create*1e3:.1f} ms, HBM {r.device_bytes()/1e6:.1f} MB, bvh {r.bvh_info()}", flush=True)
    if os.environ.get("FLUX_MATH") == "strict":   # (scripts/valu_classes.sh on the STRICT kernels)
        r.set_math(flux_amd.MATH_STRICT)
    for v in variants:
        r.set_kernel(v)
        for rep in range(2):
            t = time.time()
            img = r.render_frame()
            wall = time.time() - t
            ms = r.last_kernel_ms()
            print(f"   variant {v} rep {rep}: kernel {ms:.2f} ms  wall {wall*1e3:.1f} ms  {W*H*n*n/ms/1e3:.1f} Msamples/s  mean={img.mean():.6f}", flush=True)
    if want_stats:
        r.enable_stats(True); r.stats(reset=True); r.render_frame(); st = r.stats()
        seg = max(st["segments"], 1)
        print("   stats:", st, f"| per segment: nodes {st['bvh_nodes']/seg:.2f} tris {st['tris_tested']/seg:.2f}; "
              f"segments/sample {seg/st['samples']:.3f} matte/sample {st['matte_bounces']/st['samples']:.3f}", flush=True)
    r.close()
