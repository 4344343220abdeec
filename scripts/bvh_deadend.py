import sys, os
sys.path.insert(0, os.getcwd())
import flux_amd
from flux_amd.procedural import heightfield_scene
sd = heightfield_scene(1000, 500)
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(16, 5, 50), seed=1)
r.enable_stats(True); r.stats(reset=True); r.render_frame()
raw = r.stats_raw(); st = r.stats()
print("nodes", st["bvh_nodes"], "segments", st["segments"], "tris", st["tris_tested"], "raw10-12", raw[10:13])
print("visits per segment %.2f  dead-end visits %.1f %%  of them reached by a pop %.1f %% of all visits" % (st["bvh_nodes"]/st["segments"], 100*raw[10]/raw[11], 100*raw[12]/raw[11]))
