#!/usr/bin/env python3
"""Context creation of the 1 M-triangle scene (config 5), by phase (flux_ctx_create_timing) -- BVH build included under `host`.
usage: python scripts/hf_create_timing.py [nx nz root]     (FLUX_BUILD_THREADS=1 for the serial builder)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux_amd
from flux_amd.procedural import heightfield_scene
nx, nz, root = (int(x) for x in (sys.argv[1:4] + ["1000", "500", "64"][len(sys.argv) - 1:]))
t0 = time.perf_counter(); sd = heightfield_scene(nx, nz); t_gen = time.perf_counter() - t0
for k in range(3):
    t0 = time.perf_counter()
    r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(root, 5, 50), seed=1)
    wall = time.perf_counter() - t0
    t = r.create_timing(); b = r.bvh_info()
    print(f"threads {os.environ.get('FLUX_BUILD_THREADS', 'default')} run {k}: python wall {wall * 1e3:.1f} ms (scene generation {t_gen * 1e3:.0f} ms before it); "
          f"flux_ctx_create {t['total']:.1f} ms = host {t['host']:.1f} (binary SAH build {b['build_us'] / 1e3:.1f}) + runtime {t['runtime']:.1f} + alloc {t['alloc']:.1f} "
          f"+ upload {t['upload']:.1f} + tables {t['tables']:.1f} + free {t['free']:.1f} + other {t['other']:.1f}", flush=True)
    r.close()
