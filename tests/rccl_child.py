"""Child process of tests/test_gpu_rccl.py: ONE rank of a world-size-1 NCCL (= RCCL on ROCm) process group on cuda:0.

Started as a fresh interpreter (never exec'ed from a process that holds the GPU).  It proves what a one-GPU box can prove
about the N>1 path of bench.py / flux_amd/dist.py (the reference's fan-out + ImageBuilder gather,
fluxcore/src/manager.rs:156-162, 316-324): librccl loads BESIDE flux_amd/libflux_hip.so under the "torch first, one HIP
runtime" arrangement of flux_amd/_lib.py, a communicator comes up on the device, and all_gather_into_tensor / all_reduce /
barrier accept the sharders' f64 buffers on the stream the render kernel was launched on.

usage: rccl_child.py <out_dir> <width> <height> <sample_root> <seed>
Writes <out_dir>/sets.npy, rows.npy (the frames assembled AFTER the collective) and report.json.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, width, height, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import numpy as np
    import torch
    import torch.distributed as dist

    import flux_amd
    from conftest import small_scene
    from flux_amd.dist import FrameSharder, SetSharder, hip_render_fn, hip_render_sets_fn

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    report = {"backend": dist.get_backend(), "world": dist.get_world_size(),
              "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()),
              "hip": torch.version.hip, "device": torch.cuda.get_device_name(0)}

    sd = small_scene(flux_amd.load_scene(os.path.join(ROOT, "scenes", "demo2.yml")), width, height)
    cfg = flux_amd.JobConfiguration(n, 5, 50)
    with flux_amd.Renderer(sd, cfg, seed=seed, device=0, set_share=(0, 1)) as r:
        # --- sample-set tiles (bench.py's default sharding)
        sh = SetSharder(height, width, 0, 1, dev, torch.from_numpy(r.row_perm_table()))
        sh.render(hip_render_sets_fn(r))
        sh.collect()  # skips the collective at world 1 -> call it explicitly on the SAME buffers, as world > 1 would
        dist.all_gather_into_tensor(sh.gathered.view(-1), sh.local.view(-1))
        torch.cuda.synchronize()
        report["sets_gather_equals_local"] = bool(torch.equal(sh.gathered[0], sh.local))
        src = sh.gathered  # assemble from what came back through RCCL
        frame_sets = src[sh._g, sh._r, sh._m].cpu().numpy()
        # --- row tiles (the reference's WorkUnit rows)
        fs = FrameSharder(height, width, 0, 1, dev)
        fs.render(hip_render_fn(r))
        dist.all_gather_into_tensor(fs.gathered.view(-1), fs.local.view(-1))
        torch.cuda.synchronize()
        report["rows_gather_equals_local"] = bool(torch.equal(fs.gathered[0], fs.local))
        frame_rows = fs.gathered.permute(1, 0, 2, 3).reshape(fs.rmax, width, 3)[:height].cpu().numpy()
        # --- the other collectives bench.py issues: max-over-ranks timing, summed statistics, barrier
        t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.barrier()
        torch.cuda.synchronize()
        report["all_reduce_ok"] = bool(t[0].item() == 1.5 and t[1].item() == 2.5)
    np.save(os.path.join(out_dir, "sets.npy"), frame_sets)
    np.save(os.path.join(out_dir, "rows.npy"), frame_rows)
    # which native libraries this process really holds
    with open("/proc/self/maps") as f:
        maps = f.read()
    libs = sorted({line.split()[-1] for line in maps.splitlines() if ".so" in line})
    report["librccl"] = [p for p in libs if "rccl" in os.path.basename(p)]
    report["libflux_hip"] = [p for p in libs if "libflux_hip" in p]
    report["libamdhip64"] = [p for p in libs if "libamdhip64" in p]
    dist.destroy_process_group()
    with open(os.path.join(out_dir, "report.json"), "w") as f:
        json.dump(report, f)
    print("RCCL_CHILD " + json.dumps(report), flush=True)


if __name__ == "__main__":
    main()
