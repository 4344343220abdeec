#!/usr/bin/env python3
"""bench.py -- Msamples/s of the render loop on scenes/demo2.yml (BASELINE.json's metric).

A step = one pass of the hot path over one frame: every rank renders its rows of the 800x600 image
at sample_root^2 spp into HBM, then (N>1) one RCCL all_gather assembles the frame.  Total work is
fixed as N grows ("strong" scaling: the image is tiled across the GPUs, as the north star states).
Sample tables and the scene are resident in HBM before the timed region (they are the inputs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--root 128] [--scene demo2]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# the reference's one published run of this exact metric and config (demo2.yml, 800x600,
# 16384 spp): 1479.9 s on "44 cores" = 5.314 Msamples/s (its timer also spans scene + sample-table construction;
# the comparable span here is `reference_equivalent_s`)
PUBLISHED_MSAMPLES_S = 7864.32 / 1479.900397
# lane-operations one ray segment of demo2 needs at the least (12 sphere tests x 11 + 1.5 exact
# candidates x 30 + plane 15 + shading 50) -- the yardstick `useful_valu_frac` holds the issued lane slots against
USEFUL_LANE_OPS_PER_SEGMENT = 242.0
# Mean issue cost of a kernel's VALU instructions in shader cycles, weighted by its DYNAMIC instruction classes (hardware class
# counters, profiles/r04_experiments/valu_classes_r04.log) with the per-class costs measured in round 5
# (scripts/micro/valu_issue.hip, profiles/r05_valu_issue.json: f64 / packed f32 / three-operand integer / e64 selects 4.1,
# compares and conversions 4.7, plain f32 / integer / register moves / e32 selects 2.4, f64 transcendental seeds 16.1).  Until
# round 4 every instruction was priced at 4.  Split kernel: 422 f64 x 4.13 + 11.5 x 16.1 + 65 packed x 4.13 + 14 f32 x 2.45 +
# 15.5 cvt x 4.67 + 106 int32 x 3.3 + 402 unclassified (170 moves x 2.45, 100 compares x 4.75, 90 selects x 3.3, 42 lane ops and
# others x 4.5) = 4 040 cycles per 1 039 instructions.  The mesh kernel's node step is all 4-cycle classes.
ISSUE_CYCLES_PER_INST = {"render_split_kernel": 3.89, "render_bvh4_kernel": 4.1}
NOMINAL_CLOCK_HZ = 2.4e9  # measured for the render kernels: 2.38 GHz (scripts/kernel_clock.sh); the microbenchmark's pure fma streams: 2.16


# BASELINE.md section 5 configurations -> (scene, sample_root): 2 = demo1 @256 spp, 3 = demo2 @1024 spp, 4 = the headline
# (demo2 @16384 spp: the workload at every N, it fits one GPU), 5 = the procedural 1M-triangle height field @4096 spp.
# (1 = demo1 @16 spp on the reference's CPU path: the cpu_baseline leg.)
CONFIGS = {2: ("demo1", 16), 3: ("demo2", 32), 4: ("demo2", 128), 5: ("hf:1000x500", 64)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="a BASELINE.json configuration by number (sets --scene/--root): 2 demo1@256spp, 3 demo2@1024spp, "
                         "4 demo2@16384spp (the default headline), 5 procedural 1M-triangle height field@4096spp")
    ap.add_argument("--scene", default=None, help="demo1 | demo2 | hf:NXxNZ (procedural height field)")
    ap.add_argument("--root", type=int, default=None, help="sample_root (spp = root^2); 128 = 16384 spp")
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kernel", type=int, default=0, help="0 default, 1 static, 2 refill, 3 split")
    ap.add_argument("--math", default="fast", choices=["fast", "strict"],
                    help="render arithmetic (include/flux_abi.h FLUX_MATH_*); both are FP64 and parity-tested")
    ap.add_argument("--shard", default="auto", choices=["auto", "rows", "sets"],
                    help="how the frame is split over GPUs: interleaved rows, or sample sets (default)")
    ap.add_argument("--abi-multi", action="store_true",
                    help="ONE process driving --gpus N devices through the C ABI's multi-GPU entry (flux_multi_*: per-device contexts, one "
                         "launch per device, ncclAllGather from RCCL's C API) instead of N torch.distributed ranks: what a compiled "
                         "embedder (the reference's Rust host, the C++ `flux --split sets`) runs.  Prints the same kind of line.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rccl-probe", action="store_true",
                    help="skip the world-size-1 RCCL child after the timed region (profiling passes: a second GPU process under "
                         "the rocprofv3 preload would put its kernels into the counter files)")
    ap.add_argument("--cpu-root", type=int, default=None, help="sample_root of the bounded CPU-baseline sample")
    a = ap.parse_args()
    scene, root = CONFIGS[a.config if a.config is not None else 4]
    a.scene = a.scene or scene
    a.root = a.root or root
    if a.cpu_root is None:  # ~10-30 s of oracle work on 16 host threads
        a.cpu_root = 6 if a.scene.startswith("hf:") else 32
    return a


def host_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(sd, depth, seed, cpu_root):
    """The oracle (CPU restatement of the reference; kind 'port') timed on this box's host cores on a
    bounded sample: the SAME scene, full frame, at cpu_root^2 spp (cost is linear in spp)."""
    import flux_amd
    from oracle import oracle
    cores = host_cores()
    cfg = flux_amd.JobConfiguration(cpu_root, depth, 50)
    t0 = time.perf_counter()
    o = oracle.Oracle(sd, cfg, seed=seed)
    t_tables = time.perf_counter() - t0
    t0 = time.perf_counter()
    o.render_frame(threads=cores)
    dt = time.perf_counter() - t0
    W, H = sd.output_settings.image_width, sd.output_settings.image_height
    samples = W * H * cpu_root * cpu_root
    o.close()
    return {"value": round(samples / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{sd.scene_name}.yml full {W}x{H} frame at {cpu_root * cpu_root} spp (sample_root {cpu_root}), "
                      f"depth {depth}, seed {seed}: {samples / 1e6:.1f} Msamples in {dt:.2f} s on {cores} threads "
                      f"(row-parallel); table build {t_tables:.2f} s excluded, as for the GPU value"}


def cpu_single_thread(depth, seed):
    """BASELINE.json configs[0] / BASELINE.md section 3: scenes/demo1.yml at 16 spp (sample_root 4), 800x600, ONE host thread --
    Camera::render (trace.rs:53-97) on one core, as restated by the oracle (kind 'port'): 7.68 M samples, about 2 s."""
    import flux_amd
    from oracle import oracle
    sd = flux_amd.load_scene(os.path.join(ROOT, "scenes", "demo1.yml"))
    cfg = flux_amd.JobConfiguration(4, depth, 50)
    o = oracle.Oracle(sd, cfg, seed=seed)
    t0 = time.perf_counter()
    o.render_frame(threads=1)
    dt = time.perf_counter() - t0
    o.close()
    W, H = sd.output_settings.image_width, sd.output_settings.image_height
    samples = W * H * 16
    return {"value": round(samples / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"demo1.yml full {W}x{H} frame at 16 spp (sample_root 4), depth {depth}, seed {seed}: "
                      f"{samples / 1e6:.2f} Msamples in {dt:.2f} s on 1 thread (BASELINE.json configs[0])"}


def load_profile(scene_label, kernel_name):
    """The latest committed rocprofv3 PMC summary (profiles/*pmc*.json, scripts/summarize_profile.py) of this SCENE
    and KERNEL (any spp: the counters scale with the sample count, so they are carried per sample)."""
    import glob
    # newest = highest tag (r02f > r02e > r01j): file times mean nothing in a fresh checkout
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json")), key=os.path.basename, reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        if d.get("scene") == scene_label and d.get("kernel") == kernel_name and d.get("samples_per_launch"):
            d["file"] = os.path.basename(path)
            return d
    return None


def rccl_probe(frame_bytes):
    """N = 1 only, after the timed region: a FRESH child process (tests/rccl_child.py, the test's own child; never an exec
    from this GPU-holding process) brings up a world-size-1 NCCL (= RCCL) group on this GPU, renders a small demo2 frame
    through both sharders and runs the frame's one collective, all_gather_into_tensor, on their device buffers -- the part of
    the N > 1 path (fluxcore/src/manager.rs:316-324's gather) a single GPU can execute.  Bounded by a timeout and never
    raises: the line reports what happened."""
    import socket
    import subprocess
    import tempfile
    rep = {"ran": False}
    try:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), tmp, "64", "48", "8", "7"], env=env,
                               capture_output=True, text=True, timeout=180)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"ran": False, "error": (p.stderr or p.stdout)[-300:]}
            with open(os.path.join(tmp, "report.json")) as f:
                child = json.load(f)
        rep = {"ran": True, "backend": child["backend"], "world": child["world"], "rccl_version": child["rccl_version"],
               "all_gather_equals_local": bool(child["sets_gather_equals_local"] and child["rows_gather_equals_local"]),
               "all_reduce_ok": child["all_reduce_ok"], "one_hip_runtime": len(child["libamdhip64"]) == 1,
               "librccl": child["librccl"], "child_wall_s": round(dt, 2),
               "note": "world-size-1 group in a fresh child on this GPU, outside the timed region: library loading beside "
                       "libflux_hip.so, communicator, and the collectives on the sharders' f64 buffers; says nothing about xGMI"}
    except Exception as ex:  # noqa: BLE001 -- a probe: report, never fail the bench
        rep = {"ran": False, "error": f"{type(ex).__name__}: {ex}"[:300]}
    return rep


def multi_abi_probe(flux_amd, sd, cfg, a, frame):
    """N = 1 only, after the timed region: the SAME frame through the C ABI's multi-GPU entry (flux_multi_*: per-device context,
    one launch, ncclCommInitAll + ncclAllGather from RCCL's C API, reassembly kernel) on this one device -- what a compiled
    embedder (the reference's Rust host, the C++ `flux --split sets`) runs instead of torch.distributed.  Reports its own timing
    words and whether the frame equals the timed one bit for bit.  Never raises."""
    # RCCL prints a version banner to STDOUT when its first communicator comes up; stdout carries the ONE JSON line, so file
    # descriptor 1 points at stderr while the probe runs
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        import numpy as np
        t0 = time.perf_counter()
        with flux_amd.MultiRenderer(sd, cfg, seed=a.seed, devices=[0]) as m:
            m.set_kernel(a.kernel)
            m.set_math(flux_amd.MATH_FAST if a.math == "fast" else flux_amd.MATH_STRICT)
            t_create = time.perf_counter() - t0
            m.render_frame()                      # warm-up
            got = m.render_frame()
            rep = {"ran": True, **{k: round(v, 3) for k, v in m.timing().items()}, **m.info(),
                   "python_create_wall_ms": round(t_create * 1e3, 2),
                   "frame_equals_timed_frame": bool(np.array_equal(got, frame.cpu().numpy())),
                   "note": "flux_multi_create / flux_multi_render_frame on devices [0]: RCCL through the C ABI at G = 1 (says nothing about xGMI)"}
        flux_amd.release_comms()
        return rep
    except Exception as ex:  # noqa: BLE001 -- a probe: report, never fail the bench
        return {"ran": False, "error": f"{type(ex).__name__}: {ex}"[:300]}
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def self_launch(n_gpus):
    """`python bench.py --gpus N` from a bare shell: start N fresh ranks under torch.distributed.run and return
    their exit code.  Runs BEFORE torch / flux_amd are imported, so this process never touches the GPU (a process
    that has initialised HIP must not exec or fork GPU work); the children are new interpreters.  The reference's
    equivalent fan-out is in-process (fluxcore/src/manager.rs:156-162: one clone of the job per worker)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    # every rank appends the phases it reaches to <dir>/rank<r>.log (phase()): a hang or a dead rank is then reported as
    # "rank 5 stopped after `ctx created`", never as a silent timeout.  The child is bounded; it is a fresh process, never an exec.
    import tempfile
    limit = float(os.environ.get("FLUX_BENCH_LAUNCH_TIMEOUT_S", "900"))
    with tempfile.TemporaryDirectory(prefix="flux_bench_phases_") as pdir:
        env["FLUX_BENCH_PHASE_DIR"] = pdir
        # its own session: on a timeout the WHOLE group goes (the launcher and every rank it started), by exact process group id
        import signal
        child = subprocess.Popen(cmd, env=env, start_new_session=True)
        try:
            rc = child.wait(timeout=limit)
            timed_out = False
        except subprocess.TimeoutExpired:
            rc, timed_out = 124, True
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(child.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    child.wait(timeout=10)
                    break
                except subprocess.TimeoutExpired:
                    continue
        if rc != 0:
            print(f"bench.py: the {n_gpus}-rank run {'exceeded ' + str(limit) + ' s' if timed_out else 'exited with code ' + str(rc)}; "
                  "last phase each rank reached:", file=sys.stderr)
            for rk in range(n_gpus):
                try:
                    with open(os.path.join(pdir, f"rank{rk}.log")) as f:
                        lines = [ln.strip() for ln in f if ln.strip()]
                except OSError:
                    lines = []
                print(f"  rank {rk}: {lines[-1] if lines else 'never started (no phase recorded)'}", file=sys.stderr)
    return rc


_T_START = time.perf_counter()


def phase(rank, world, what):
    """One line per phase a rank reaches (imported / group up / ctx created / warm-up done / timed / statistics / done): to
    <FLUX_BENCH_PHASE_DIR>/rank<r>.log when self_launch set it, and to stderr when there is more than one rank (stdout carries
    the ONE JSON line only)."""
    line = f"{time.perf_counter() - _T_START:8.2f} s  {what}"
    d = os.environ.get("FLUX_BENCH_PHASE_DIR")
    if d:
        try:
            with open(os.path.join(d, f"rank{rank}.log"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass
    if world > 1:
        print(f"bench.py rank {rank}/{world}: {line}", file=sys.stderr, flush=True)


def abi_multi_main(a):
    """`bench.py --abi-multi --gpus N`: the frame of every step comes from flux_multi_render_frame_device (left in HBM on device 0)."""
    import flux_amd
    sd = flux_amd.load_scene(os.path.join(ROOT, "scenes", f"{a.scene}.yml"))
    W, H = sd.output_settings.image_width, sd.output_settings.image_height
    n = a.root
    cfg = flux_amd.JobConfiguration(n, a.depth, 50)
    shard = {"auto": flux_amd.SHARD_AUTO, "sets": flux_amd.SHARD_SETS, "rows": flux_amd.SHARD_ROWS}[a.shard]
    sys.stdout.flush()
    saved = os.dup(1)   # RCCL's banner goes to stdout: keep it off the ONE JSON line
    os.dup2(2, 1)
    try:
        t0 = time.perf_counter()
        m = flux_amd.MultiRenderer(sd, cfg, seed=a.seed, devices=list(range(a.gpus)), shard=shard)
        t_create = time.perf_counter() - t0
        m.set_kernel(a.kernel)
        m.set_math(flux_amd.MATH_FAST if a.math == "fast" else flux_amd.MATH_STRICT)
        for _ in range(a.warmup):
            m.render_frame_device()
        spans = []
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m.render_frame_device()
            spans.append(m.timing())
        elapsed = time.perf_counter() - t0
        frame = m.render_frame()
        info = m.info()
        plan = m.rank_renderer(0).launch_plan(num_sets=len(range(0, W, a.gpus))) if info["shard"] == flux_amd.SHARD_SETS else \
            m.rank_renderer(0).launch_plan(num_rows=len(range(0, H, a.gpus)))
        m.close()
        flux_amd.release_comms()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    import numpy as np
    samples = W * H * n * n
    mean = {k: sum(s[k] for s in spans) / max(len(spans), 1) for k in spans[0]}
    out = {"metric": f"Msamples/sec on {a.scene}.yml (fixed spp)", "value": round(samples * a.steps / elapsed / 1e6, 3), "unit": "Msamples/s",
           "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"scenes/{a.scene}.yml {W}x{H} at {n * n} spp (sample_root {n}), depth {a.depth}, seed {a.seed}",
                      "parallelism": f"C ABI flux_multi_*: ONE process, {a.gpus} device(s), "
                                     f"{'pixel-set' if info['shard'] == flux_amd.SHARD_SETS else 'row-interleaved'} tiles, 1 ncclAllGather (RCCL "
                                     f"{info['rccl_version']}), reassembly on device 0", "math": a.math, "finite": bool(np.isfinite(frame).all()),
                      "waves_per_pixel": plan["waves_per_pixel"], "blocks_per_launch": plan["blocks"]},
           "step_breakdown_ms": {"render": round(mean["kernel_ms"], 3), "all_gather": round(mean["all_gather_ms"], 3),
                                 "reassembly": round(mean["reassembly_ms"], 3), "frame_call": round(mean["frame_ms"], 3),
                                 "note": "flux_multi_timing: the slowest rank's kernel (HIP events on its stream), the all-gather and the "
                                         "reassembly on device 0's stream, wall time of the render call"},
           "multi_create_ms": round(t_create * 1e3, 1), "multi_info": info,
           "build_id": (flux_amd._lib.lib.flux_build_id() or b"").decode(),
           "roofline": None, "cpu_baseline": None,
           "note": "the C-ABI multi-GPU path; the roofline / cpu_baseline objects are in the default (torch.distributed) line"}
    print(json.dumps(out), flush=True)


def main():
    a = parse()
    if a.abi_multi:
        return abi_multi_main(a)
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but launched with WORLD_SIZE {world}; using {world}", file=sys.stderr)
        a.gpus = world

    import torch
    import torch.distributed as dist

    import flux_amd
    from flux_amd.dist import FrameSharder, SetSharder, hip_render_fn, hip_render_sets_fn

    phase(rank, world, "imported torch + flux_amd")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the renderer has no CPU fallback)")
    # Rehearsal of the N > 1 path on a one-GPU box (FLUX_BENCH_REHEARSE=1): every rank on device 0, the gather over gloo
    # through the host -- everything but RCCL itself; never the measured configuration (the line says so)
    rehearse = os.environ.get("FLUX_BENCH_REHEARSE") == "1" and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # bounded: a rank that never arrives ends the job with an error after two minutes instead of hanging it
        from datetime import timedelta
        limit = timedelta(seconds=float(os.environ.get("FLUX_BENCH_GROUP_TIMEOUT_S", "120")))
        if rehearse:
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        phase(rank, world, f"process group up ({dist.get_backend()})")

    if a.scene.startswith("hf:"):  # BASELINE config 5: procedural height field, e.g. hf:1000x500 = 1M triangles
        from flux_amd.procedural import heightfield_scene
        nx, nz = [int(x) for x in a.scene[3:].split("x")]
        sd = heightfield_scene(nx, nz)
        scene_label = f"procedural height field {nx}x{nz} ({2 * nx * nz} triangles, flux_amd/procedural.py) in the demo2 set"
    else:
        sd = flux_amd.load_scene(os.path.join(ROOT, "scenes", f"{a.scene}.yml"))
        scene_label = f"scenes/{a.scene}.yml"
    W, H = sd.output_settings.image_width, sd.output_settings.image_height
    n = a.root
    cfg = flux_amd.JobConfiguration(n, a.depth, 50)

    # Shard by sample set (one pixel per row per owned set: balanced, and each rank keeps the full-frame table
    # locality, flux_amd/dist.py SetSharder) -- the same code path at every N; `--shard rows` forces row tiles.
    # A set-sharded rank builds and holds only its own sets' tables (flux_ctx_create_sets).
    use_sets = a.shard == "sets" or (a.shard == "auto" and n * n >= 64)
    share = (rank, world) if use_sets else None
    # The FIRST context of the process: its wall time contains the HIP runtime's lazy initialisation (device context, copy engine,
    # compute queue, code object: scripts/micro/cold_start.hip) -- process start-up, which the reference's job timer never sees.
    t0 = time.perf_counter()
    r = flux_amd.Renderer(sd, cfg, seed=a.seed, device=local_rank, set_share=share)
    torch.cuda.synchronize()
    t_create_cold = time.perf_counter() - t0
    create_cold = r.create_timing()
    # ... and a SECOND, identical context beside it: what Scene::from_data + Camera::new (workers.rs:46-54) cost a running worker,
    # i.e. what the reference's timer (manager.rs:145 -> 170) spans for every job.  Created, timed, destroyed.
    t0 = time.perf_counter()
    r2 = flux_amd.Renderer(sd, cfg, seed=a.seed, device=local_rank, set_share=share)
    torch.cuda.synchronize()
    t_create = time.perf_counter() - t0
    create_warm = r2.create_timing()
    r2.close()
    del r2
    phase(rank, world, f"ctx created ({t_create_cold * 1e3:.0f} ms the process's first, {t_create * 1e3:.1f} ms a second one)")
    r.set_kernel(a.kernel)
    r.set_math(flux_amd.MATH_FAST if a.math == "fast" else flux_amd.MATH_STRICT)
    if use_sets:
        sh = SetSharder(H, W, rank, world, dev, torch.from_numpy(r.row_perm_table()))
        fn = hip_render_sets_fn(r)
    else:
        sh = FrameSharder(H, W, rank, world, dev)
        fn = hip_render_fn(r)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    frame = None
    for _ in range(a.warmup):
        frame = sh.step(fn)
    barrier()
    phase(rank, world, f"warm-up done ({a.warmup} steps)")
    # HIP events on the stream the kernel is launched on (torch's current stream: hip_render_*_fn passes it down, and
    # there is ONE HIP runtime in the process, flux_amd/_lib.py): render | all_gather | reassembly, per step
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.steps)]
    t0 = time.perf_counter()
    for k in range(a.steps):
        ev[k][0].record()
        sh.render(fn)
        ev[k][1].record()
        sh.collect()
        ev[k][2].record()
        frame = sh.assemble()
        ev[k][3].record()
    barrier()
    elapsed = time.perf_counter() - t0
    phases = [sum(e[j].elapsed_time(e[j + 1]) for e in ev) / max(a.steps, 1) for j in range(3)]

    phase(rank, world, f"timed {a.steps} steps ({elapsed / max(a.steps, 1) * 1e3:.2f} ms per step here)")

    # every rank's own numbers (manager.rs:145,156-170: the reference's timer spans the slowest worker; here each rank's share is
    # kept so that a slow rank, a slow gather or a slow Python launch path can be told apart): one all_gather of 5 doubles
    t = torch.tensor([elapsed] + phases + [elapsed / max(a.steps, 1) * 1e3 - sum(phases)], dtype=torch.float64, device=dev)
    if world > 1:
        from flux_amd.dist import _all_gather
        allt = torch.empty((world, t.numel()), dtype=torch.float64, device=dev)
        _all_gather(allt, t)  # (over gloo -- the one-GPU rehearsal -- through the host)
    else:
        allt = t.reshape(1, -1)
    allt = allt.cpu()
    elapsed_max, kernel_ms_max, gather_ms_max, assemble_ms_max, overhead_ms_max = [float(x) for x in allt.max(dim=0).values]
    per_rank = [{"rank": k, "wall_ms_per_step": round(float(allt[k, 0]) / max(a.steps, 1) * 1e3, 3), "render": round(float(allt[k, 1]), 3),
                 "all_gather": round(float(allt[k, 2]), 3), "reassembly": round(float(allt[k, 3]), 3),
                 "launch_overhead_ms": round(float(allt[k, 4]), 3)} for k in range(world)]
    spread = {name: {"min": round(float(allt[:, j].min()), 3), "mean": round(float(allt[:, j].mean()), 3),
                     "max": round(float(allt[:, j].max()), 3)}
              for j, name in ((1, "render"), (2, "all_gather"), (3, "reassembly"), (4, "launch_overhead_ms"))}

    # the frame's way to the host (the reference-equivalent span ends with the image in host memory, manager.rs:316-324): one
    # device-to-host copy of the assembled frame into pageable memory, timed on rank 0 (untimed extra)
    t0 = time.perf_counter()
    frame_host = frame.cpu()
    t_d2h = time.perf_counter() - t0
    del frame_host

    # what HBM delivers on THIS device ("report against both" the 8 TB/s spec and a measured copy): a 1 GiB
    # device-to-device copy, read + write bytes, best of 5 (untimed extra)
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e30
        for _ in range(5):
            e0.record()
            dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        copy_gbs = 2.0 * src.numel() * 4 / (best * 1e-3) / 1e9
        del src, dst

    # exact path statistics of THIS rank's share (untimed extra pass) -> algorithmic bytes
    r.enable_stats(True)
    r.stats(reset=True)
    sh.render(fn)
    torch.cuda.synchronize()
    st = r.stats(reset=True)
    r.enable_stats(False)
    phase(rank, world, "statistics pass done")
    stt = torch.tensor([st["samples"], st["matte_bounces"], st["segments"], st["glossy_bounces"], st["bvh_nodes"],
                        st["tris_tested"], st["misses"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(stt, op=dist.ReduceOp.SUM)
    tot_samples, tot_matte, tot_segments, tot_glossy, tot_nodes, tot_tris, tot_miss = [float(x) for x in stt]
    bvh = r.bvh_info()
    # the kernel the timed call launched: asked of the library's launch planner for exactly that call (never re-derived here)
    plan = r.launch_plan(num_sets=sh.count) if use_sets else r.launch_plan(num_rows=sh.count)
    kernel_name = {flux_amd._lib.PLAN_STATIC: "render_static_kernel", flux_amd._lib.PLAN_REFILL: "render_refill_kernel",
                   flux_amd._lib.PLAN_SPLIT: "render_split_kernel", flux_amd._lib.PLAN_BVH_BINARY: "render_bvh_kernel",
                   flux_amd._lib.PLAN_BVH4: "render_bvh4_kernel"}.get(plan["kernel"], "none")

    if rank == 0:
        samples = W * H * n * n
        assert int(tot_samples) == samples, (tot_samples, samples)
        finite = bool(torch.isfinite(frame).all())
        mbar = tot_matte / tot_samples
        gbar = tot_glossy / tot_samples
        # pixel 16 B + lens 16 B + 24 B per Matte bounce, plus (triangle scenes) the BVH nodes visited
        # and triangles tested at their laid-out sizes (64 B / 128 B).  (As laid out here a glossy bounce also reads 24 B
        # of tabulated lobe factors: reported beside it, not counted in `achieved`.)
        table_bytes = 32.0 + 24.0 * mbar
        # as laid out for the kernel that ran: 64-B 4-wide nodes and 128-B leaf records holding one or two triangles (counted
        # per triangle test at 64 B when the record is a quad's) for render_bvh4_kernel; 32-B nodes (two 16-B gathers of a
        # DevNodeQ) + 80 B of a DevTri for the binary-tree kernel; the inline walk of the other kernels reads DevNode / DevTri
        if plan["kernel"] == flux_amd._lib.PLAN_BVH4:
            tri_b = bvh["leaf_record_bytes"] * bvh["leaf_records"] / max(bvh["triangles"], 1)
            bvh_bytes = (tot_nodes * bvh["wide_node_bytes"] + tot_tris * tri_b) / tot_samples
        else:
            bvh_bytes = (tot_nodes * bvh["node_bytes"] + tot_tris * bvh["tri_bytes"]) / tot_samples
        bytes_per_sample = table_bytes + bvh_bytes
        # the dominant kernel's launch on rank 0 covers samples/world camera paths + its share of the framebuffer
        samples_launch = samples / world
        alg_bytes_launch = samples_launch * bytes_per_sample + (H * W / world) * 24.0
        alg_gbs = alg_bytes_launch / (kernel_ms_max * 1e-3) / 1e9
        workload = f"{scene_label} {W}x{H} at {n * n} spp (sample_root {n}), depth {a.depth}, seed {a.seed}"
        prof = load_profile(scene_label, kernel_name)
        # counters of the committed rocprofv3 PMC passes of this scene + kernel, carried per sample (NOT measured in this
        # run: `from_committed_profile`)
        scale = samples_launch / prof["samples_per_launch"] if prof else None
        l2_miss = None
        if prof:
            l2_miss = prof.get("l2_miss_bytes_per_launch") or prof.get("hbm_bytes_per_launch")  # (the key's name before round 3)
        traffic = l2_miss * scale if l2_miss else None
        traffic_gbs = traffic / (kernel_ms_max * 1e-3) / 1e9 if traffic else None
        # ONE basis for `achieved` / `frac`: the algorithmic bytes of SURVEY.md 8(d).  Where they exceed what HBM could
        # deliver (config 5: every node / triangle record a lane reads, served by L1 / L2 / Infinity Cache) no HBM fraction is
        # claimed: frac is null.  The PMC figure is published beside it for what it is -- the bytes that left the L2s towards
        # the fabric (FETCH_SIZE x 2 + WRITE_SIZE: one 128-B line per miss, tallied at 64 B, also for divergent 16-B gathers;
        # Infinity-Cache hits INCLUDED: profiles/r03_gather_calibration.json), an upper bound on HBM bytes.  gfx950 exposes no
        # DRAM-side counter (profiles/r03_experiments/rocprofv3_counter_names_gfx950.txt), so hbm_bytes is null.
        if alg_gbs <= HBM_PEAK_GBS:
            achieved, basis = alg_gbs, "algorithmic bytes (SURVEY.md 8d) / kernel time"
        else:
            achieved, basis = None, (f"the algorithmic figure ({alg_gbs:.0f} GB/s) exceeds the HBM peak: the records are served by "
                                     "L1 / L2 / Infinity Cache, and no counter on gfx950 separates HBM from Infinity-Cache traffic "
                                     "(l2_miss_traffic_gbs is an upper bound on the HBM rate); no HBM fraction is claimed")
        valu_insts = prof["valu_insts_per_launch"] * scale if prof and prof.get("valu_insts_per_launch") else None
        # VALU issue ceiling: 256 CU x 4 SIMD wave-instructions per (mean cycles per instruction of THIS kernel's mix) at the
        # nominal 2.4 GHz -- 614 G/s at 4 cycles each, 632 G/s for the split kernel's measured mix
        # ... from the committed profile's own class counters when it has them (scripts/valu_classes.sh + summarize_profile.py), else
        # the round-5 figures above
        issue_cycles = (prof or {}).get("issue_cycles_per_inst") or ISSUE_CYCLES_PER_INST.get(kernel_name, 4.1)
        issue_peak = 256 * 4 * NOMINAL_CLOCK_HZ / issue_cycles
        out = {
            "metric": (f"Msamples/sec on {a.scene}.yml (fixed spp)" if not a.scene.startswith("hf:") else
                       f"Msamples/sec on the procedural {a.scene[3:]} height field in the demo2 set (fixed spp)"),
            "value": round(samples * a.steps / elapsed_max / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed_max / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": (round(samples * a.steps / elapsed_max / 1e6 / PUBLISHED_MSAMPLES_S, 1)
                            if a.scene == "demo2" and n == 128 else None),
            # the one published run (README.md:1: 1479.9 s on "44 cores") is quoted for 16384 spp, but demo.png carries the per-pixel
            # variance of a 65536-spp render of this estimator (tests/test_gpu_ref16.py::test_variance_profile):
            # the published time may belong to a 4x larger job, and vs_baseline would then be 4x too high
            "published_run_note": ("vs_baseline divides by README.md's 1479.9 s for '16384 spp'; demo.png's variance matches 65536 spp "
                                   "(test_variance_profile), so that run may have been 4x this job" if a.scene == "demo2" and n == 128
                                   else None),
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "scene": scene_label, "kernel": kernel_name.replace("render_", "").replace("_kernel", ""),
                       "math": a.math, "waves_per_pixel": plan["waves_per_pixel"], "threads_per_block": plan["block"],
                       "blocks_per_launch": plan["blocks"],
                       "parallelism": (f"pixel-set tiles (one pixel per row per owned sample set; each rank holds only its "
                                       f"sets' tables) over {world} GPU(s), 1 all_gather" if use_sets else
                                       f"row-interleaved image tiles over {world} GPU(s), 1 all_gather"),
                       "finite": finite, "backend": (dist.get_backend() if world > 1 else "none (single process)"),
                       "rccl_ranks": (dist.get_world_size() if world > 1 and not rehearse else 0),
                       **({"rehearsal": "all ranks on ONE GPU, gather over gloo: not a measurement"} if rehearse else {})},
            "step_breakdown_ms": {"render": round(kernel_ms_max, 3), "all_gather": round(gather_ms_max, 3),
                                  "reassembly": round(assemble_ms_max, 3),
                                  "launch_overhead_ms": round(overhead_ms_max, 3),
                                  "spread_over_ranks": spread, "per_rank": per_rank,
                                  "note": "HIP events on the launch stream, mean over steps; render / all_gather / reassembly at the top "
                                          "level are the max over ranks; launch_overhead_ms = a rank's wall time per step minus its three "
                                          "event spans (host-side launch path, event gaps, the barrier wait for slower ranks)"},
            "roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "basis": basis,
                         "traffic_is": ("L2-miss bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, rocprofv3 PMC, separate passes): they "
                                        "include Infinity-Cache hits, so this is an UPPER bound on HBM bytes"),
                         "l2_miss_bytes": traffic, "hbm_bytes": None,
                         "hbm_bytes_note": "gfx950 / rocprofv3 7.2 expose no DRAM-side counter (TCC_EA0_RDREQ_DRAM counts the same "
                                           "requests warm and cold: profiles/r03_gather_calibration.json)",
                         "from_committed_profile": prof is not None,
                         "from_committed_profile_fields": ["traffic", "l2_miss_bytes", "l2_miss_traffic_gbs", "fp64_issue_frac",
                                                           "valu_busy_frac", "lanes_active_frac", "useful_valu_frac"],
                         "measured_copy_gbs": None if copy_gbs is None else round(copy_gbs, 1),
                         "frac_of_measured_copy": (None if achieved is None or not copy_gbs else
                                                   round(achieved / copy_gbs, 6)),
                         "algorithmic_gbs": round(alg_gbs, 3),
                         "l2_miss_traffic_gbs": None if traffic_gbs is None else round(traffic_gbs, 3),
                         "profile": prof["file"] if prof else None,
                         "kernel": kernel_name,
                         "kernel_ms": round(kernel_ms_max, 3), "samples_per_launch": samples_launch,
                         "bytes_per_sample": round(bytes_per_sample, 3),
                         "bytes_per_sample_as_laid_out": round(bytes_per_sample + 24.0 * gbar, 3),
                         # what actually bounds the analytic kernels: VALU issue (f64, packed f32, compares, e64 selects: 4+ cycles of
                         # a SIMD each; plain 32-bit arithmetic and moves 2.4) -- instruction counts from the committed PMC profile
                         "fp64_issue_frac": (None if valu_insts is None else
                                             round(valu_insts / (kernel_ms_max * 1e-3) / issue_peak, 4)),
                         "issue_cycles_per_inst": issue_cycles,
                         "issue_cycles_from": ("the committed profile's dynamic class counters" if (prof or {}).get("issue_cycles_per_inst")
                                               else "round 5's class mix (bench.py ISSUE_CYCLES_PER_INST)"),
                         "valu_classes_per_64_samples": (prof or {}).get("valu_classes_per_64_samples"),
                         "issue_note": "fp64_issue_frac = VALU instructions/s x mean cycles per instruction of this kernel's class mix "
                                       "(profiles/r05_valu_issue.json) / (1024 SIMDs x 2.4 GHz; the render kernels run at 2.38 GHz, "
                                       "scripts/kernel_clock.sh): the fraction of cycles in which the VALU issues.  valu_busy_frac (SQ_ACTIVE_INST_VALU x 4 "
                                       "/ cycles) charges every plain instruction 4 cycles and overstates by the two-cycle share",
                         "valu_busy_frac": prof.get("valu_busy_frac") if prof else None,
                         "lanes_active_frac": prof.get("lanes_active_frac") if prof else None,
                         "useful_valu_frac": (None if valu_insts is None or bvh["triangles"] or a.scene != "demo2" else
                                              round(USEFUL_LANE_OPS_PER_SEGMENT * tot_segments / world / (64.0 * valu_insts), 4)),
                         "matte_bounces_per_sample": round(mbar, 5),
                         "segments_per_sample": round(tot_segments / tot_samples, 5),
                         "glossy_bounces_per_sample": round(gbar, 5),
                         "bvh_nodes_per_sample": round(tot_nodes / tot_samples, 3),
                         "tris_tested_per_sample": round(tot_tris / tot_samples, 3),
                         "misses": int(tot_miss),
                         "note": "FP64 path tracer: analytic shapes live in SGPRs; the sample tables (and, for triangle "
                                 "scenes, BVH nodes/triangles) are the only streamed data; bytes are the algorithmic figure, "
                                 "not inflated; with the set-grouped pixel order nearly all table bytes are served by the XCD "
                                 "L2s (`traffic`: what reached the memory side).  The analytic kernels are VALU-issue bound "
                                 "(valu_busy_frac); lanes_active_frac = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU); "
                                 "useful_valu_frac = model lane-operations (DESIGN.md section 4) / issued lane slots"},
            # the job as the reference times it (manager.rs:145 -> 170: Scene::from_data + Camera::new incl. MasterSampleSets::new,
            # workers.rs:46-54, then the render, then the rows in host memory): a running worker's context creation (the SECOND
            # context of this process) + one frame + the frame's copy to the host.  `_cold_s`: the same with the process's FIRST
            # context, whose creation also pays the HIP runtime's one-time initialisation.
            "ctx_create_ms": round(t_create * 1e3, 2),
            "ctx_create_breakdown_ms": {k: round(v, 3) for k, v in create_warm.items()},
            "ctx_create_cold_ms": round(t_create_cold * 1e3, 1),
            "ctx_create_cold_breakdown_ms": {k: round(v, 3) for k, v in create_cold.items()},
            "ctx_create_note": "flux_ctx_create_timing: host = validation + scene records (+ BVH build), runtime = HIP device initialisation, "
                               "alloc / upload / tables (MasterSampleSets::new on the device) / free / other; cold = the process's first "
                               "context (runtime + first-copy / first-launch set-up land in it), ctx_create_ms = a second context",
            "frame_d2h_ms": round(t_d2h * 1e3, 3),
            "reference_equivalent_s": round(t_create + elapsed_max / a.steps + t_d2h, 4),
            "reference_equivalent_cold_s": round(t_create_cold + elapsed_max / a.steps + t_d2h, 4),
            "build_id": (flux_amd._lib.lib.flux_build_id() or b"").decode(),
        }
        out["reference_equivalent_msamples_s"] = round(samples / out["reference_equivalent_s"] / 1e6, 1)
        if out["vs_baseline"] is not None:
            # the published 1479.9 s spans table construction + render + gather: compared on the same span
            out["vs_baseline"] = round(samples / out["reference_equivalent_s"] / 1e6 / PUBLISHED_MSAMPLES_S, 1)
            out["vs_baseline_span"] = ("reference-equivalent (context creation of a running worker + one frame + the frame's copy to "
                                       "the host) against README.md's 1479.9 s, which spans the same steps; value / 5.314 would be "
                                       f"{round(samples * a.steps / elapsed_max / 1e6 / PUBLISHED_MSAMPLES_S, 1)}")
        prof_kernels = (prof or {}).get("build_id", "")
        out["roofline"]["profile_build_id"] = prof_kernels or None
        out["roofline"]["profile_head"] = (prof or {}).get("git_head")
        same = bool(prof_kernels) and prof_kernels.split("kernels:")[-1] == out["build_id"].split("kernels:")[-1]
        out["roofline"]["profile_matches_build"] = same if prof else None
        if prof and not same:
            out["roofline"]["profile_warning"] = ("the committed PMC profile was taken from another build of the kernels: the "
                                                  "from_committed_profile fields describe THAT binary")
        if world == 1 and not a.no_rccl_probe:
            out["rccl_probe"] = rccl_probe(int(sh.local.numel() * 8))
            out["multi_abi"] = multi_abi_probe(flux_amd, sd, cfg, a, frame)
        if world == 1 and not a.no_cpu_baseline:
            if a.scene.startswith("hf:"):
                # the CPU checker scans every triangle per ray (it DEFINES what the BVH must reproduce): ~5 ms per ray on
                # a 1M-triangle mesh, so no bounded sample of this workload is meaningful; the headline line carries it
                out["cpu_baseline"] = None
            else:
                out["cpu_baseline"] = cpu_baseline(sd, a.depth, a.seed, a.cpu_root)
                out["cpu_baseline"]["single_thread"] = cpu_single_thread(a.depth, a.seed)
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    phase(rank, world, "done")


if __name__ == "__main__":
    main()
