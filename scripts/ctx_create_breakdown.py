#!/usr/bin/env python3
"""Where flux_ctx_create's wall time goes (flux_ctx_create_timing), cold (the process's first context) and warm (later ones),
for the headline job (demo2 @ sample_root 128) and a set share of it.  FLUX_NO_TORCH=1 keeps torch out of the process.

    python scripts/ctx_create_breakdown.py [root] [repeats]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux_amd  # noqa: E402

root = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sd = flux_amd.load_scene(os.path.join(ROOT, "scenes", "demo2.yml"))
cfg = flux_amd.JobConfiguration(root, 5, 50)
out = {"root": root, "torch_loaded": "torch" in sys.modules, "runs": []}
for k in range(reps):
    for share in (None, (0, 8)):
        t0 = time.perf_counter()
        r = flux_amd.Renderer(sd, cfg, seed=1, device=0, set_share=share)
        wall = (time.perf_counter() - t0) * 1e3
        t = r.create_timing()
        t1 = time.perf_counter()
        r.close()
        destroy = (time.perf_counter() - t1) * 1e3
        out["runs"].append({"k": k, "share": share, "python_wall_ms": round(wall, 2), "destroy_ms": round(destroy, 2),
                            "device_bytes": None, **{a: round(b, 3) for a, b in t.items()}})
        print(json.dumps(out["runs"][-1]), flush=True)
