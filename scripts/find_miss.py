"""Debug aid: locate rays that miss everything in the enclosed height-field scene."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flux_amd
from flux_amd.procedural import heightfield_scene
sd = heightfield_scene(1000, 500)
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(32, 5, 50), seed=1)
r.enable_stats(True)
bad = []
for row in range(600):
    r.stats(reset=True)
    img = r.render_rows(row, row)
    st = r.stats()
    if st["misses"] or not np.isfinite(img).all():
        bad.append(row); print("row", row, st, "finite", np.isfinite(img).all(), flush=True)
print("bad rows", bad)
for row in bad[:2]:
    a = r.render_rows(row, row)
    r.set_traversal(1)
    r.stats(reset=True)
    b = r.render_rows(row, row)
    print("row", row, "brute stats", r.stats(), "max diff bvh vs brute", np.abs(a - b).max(), "argmax", np.unravel_index(np.abs(a-b).argmax(), a.shape), flush=True)
    r.set_traversal(0)
