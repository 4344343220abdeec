"""Kernel time of ONE rank's share of the frame for G = 1, 2, 4, 8, measured on a single GPU: what each GPU of an N-GPU run
executes, without the collective -- EVERY rank's share is rendered and the slowest one counts.  `rows`: row-interleaved
image tiles (FrameSharder, `bench.py --shard rows`: the north star's tiling and the reference's WorkUnit rows); `sets`:
sample-set tiles (SetSharder, one pixel per row per owned set).  usage: shard_time.py [root] [rows|sets]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flux_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sd = flux_amd.load_scene("scenes/demo2.yml")
W, H = sd.output_settings.image_width, sd.output_settings.image_height
r = flux_amd.Renderer(sd, flux_amd.JobConfiguration(n, 5, 50), seed=1)
out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
mode = sys.argv[2] if len(sys.argv) > 2 else "rows"   # rows | sets
base = None
for G in (1, 2, 4, 8):
    times = []
    for rank in range(G):
        cnt = (H - rank + G - 1) // G
        best = 1e30
        for _ in range(2):
            if mode == "sets":
                r.render_sets_device(rank, G, len(range(rank, W, G)), out.data_ptr(), stream)
            else:
                r.render_rows_device(rank, G, cnt, out.data_ptr(), stream)
            torch.cuda.synchronize()
            best = min(best, r.last_kernel_ms())
        times.append(best)
    t = max(times)
    base = base or t
    print(f"{mode} G={G}: slowest of {G} ranks {t:8.2f} ms (fastest {min(times):8.2f})  ideal {base / G:8.2f} ms  efficiency {base / G / t * 100:5.1f}%  "
          f"-> {W * H * n * n / t / 1e3:9.1f} Msamples/s aggregate", flush=True)
