import sys, os
sys.path.insert(0, os.getcwd())
import flux_amd
sd = flux_amd.load_scene("scenes/demo2.yml")
for n in (32, 128):
    r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
    r.set_math(flux_amd.MATH_STRICT)
    for rep in range(2):
        r.render_frame(); ms = r.last_kernel_ms()
    print(f"STRICT demo2 n={n}: kernel {ms:.2f} ms  {800*600*n*n/ms/1e3:.1f} Msamples/s plan {r.launch_plan()}", flush=True)
    r.close()
