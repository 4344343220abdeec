import sys, os
sys.path.insert(0, os.getcwd())
import flux_amd
sd = flux_amd.load_scene("scenes/demo2.yml")
for D in (1, 5):
    r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(64, D, 50), seed=1)
    r.set_kernel(3)
    r.render_frame(); r.render_frame()
    print("depth", D, "kernel ms", r.last_kernel_ms())
    r.close()
