"""Mesh kernel on a mesh whose leaves do NOT fuse into quads (a triangle soup: every two-triangle leaf of the binary tree becomes a
node with two one-triangle leaves in the arena layout, flux_bvh.h): kernel ms, visits per segment.  usage: soup_time.py [triangles] [root]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flux_amd
from flux_amd.scene import MeshData
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(7)
c = rng.uniform([-12, 0.2, -8], [12, 3.0, 18], (nt, 3))
v = (c[:, None, :] + rng.normal(scale=0.08, size=(nt, 3, 3))).reshape(-1, 3)
t = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
sd = copy.deepcopy(flux_amd.load_scene("scenes/demo2.yml"))
sd.shapes = [s for s in sd.shapes if not (isinstance(s, flux_amd.SphereData) and s.radius == 1.0)]
sd.shapes.append(MeshData(v, t, flux_amd.MatteData((0.6, 0.5, 0.4), (0, 0, 0), 0.9)))
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
print("bvh", r.bvh_info(), "plan", r.launch_plan())
for rep in range(2):
    img = r.render_frame(); ms = r.last_kernel_ms()
r.enable_stats(True); r.stats(reset=True); r.render_frame(); st = r.stats()
print(f"soup {nt} triangles n={n}: kernel {ms:.2f} ms  mean {img.mean():.6f}  nodes/segment {st['bvh_nodes']/st['segments']:.2f} tris/segment {st['tris_tested']/st['segments']:.2f}")
