"""bench.py's contract on the GPU box: one JSON line with BASELINE.json's metric, the roofline and cpu_baseline objects, the
N = 1 RCCL probe -- run once on a small configuration (BASELINE config 2: demo1 @256 spp) in a child process."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--steps", "2", "--warmup", "1", "--cpu-root", "8"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly ONE line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "Msamples/s" and d["dtype"] == "f64"
    assert d["value"] > 1000.0 and d["higher_is_better"] is True and "demo1" in d["metric"] and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "l2_miss_bytes", "hbm_bytes", "from_committed_profile"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["hbm_bytes"] is None
    assert r["achieved"] is not None and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "demo1" in c["sample"]
    one = c["single_thread"]                            # BASELINE.json configs[0]: demo1 @16 spp, one host thread
    assert one["cores"] == 1 and one["kind"] == "port" and one["value"] > 0 and "16 spp" in one["sample"]
    assert d["roofline"]["kernel"] == "render_split_kernel" and d["config"]["waves_per_pixel"] == 1
    assert d["rccl_probe"]["ran"] is True and d["rccl_probe"]["backend"] == "nccl" and d["rccl_probe"]["all_gather_equals_local"] is True
    # round 6: the same frame through the C ABI's multi-GPU entry (RCCL from C, G = 1), bit-equal to the timed frame
    m = d["multi_abi"]
    assert m["ran"] is True and m["devices"] == 1 and m["rccl_version"] >= 20000 and m["frame_equals_timed_frame"] is True
    assert m["kernel_ms"] > 0 and m["frame_ms"] >= m["kernel_ms"]
    # ... context creation as the reference's timer sees it: a running worker's (warm), the process's first (cold), the breakdown
    assert 0 < d["ctx_create_ms"] <= d["ctx_create_cold_ms"]
    b = d["ctx_create_breakdown_ms"]
    assert abs(sum(v for k, v in b.items() if k != "total") - b["total"]) < 0.01 and b["tables"] > 0
    assert d["reference_equivalent_s"] <= d["reference_equivalent_cold_s"]
    assert d["build_id"].startswith("lib:") and "kernels:" in d["build_id"]
    assert d["roofline"]["profile_matches_build"] in (True, False, None)


@pytest.mark.parametrize("shard", ["sets", "rows"])
def test_bench_multi_rank_branch_rehearsal(shard):
    """bench.py's N > 1 branch (the reference's fan-out and gather: fluxcore/src/manager.rs:156-162, 316-324), kept alive in
    the driver's test run although a test box has ONE GPU: FLUX_BENCH_REHEARSE=1 puts both ranks on device 0 and runs the
    gather over gloo through the host -- everything but RCCL itself (tests/test_gpu_rccl.py covers that at world size 1).
    The child is a FRESH interpreter: bench.py starts its ranks (torch.distributed.run) before anything touches the GPU.
    Inside, rank 0 asserts that the path statistics summed over the ranks count exactly W x H x n^2 samples."""
    env = dict(os.environ, FLUX_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "3", "--steps", "2", "--warmup", "1",
                        "--shard", shard], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints, once
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and "demo2" in d["metric"]
    c = d["config"]
    assert "rehearsal" in c and c["backend"] == "gloo" and c["rccl_ranks"] == 0   # marked: never a measurement
    assert c["finite"] is True
    assert ("pixel-set tiles" if shard == "sets" else "row-interleaved") in c["parallelism"]
    assert d["roofline"]["samples_per_launch"] == 800 * 600 * 1024 / 2            # each rank's launch covers half the frame
    assert d["roofline"]["kernel"] == "render_split_kernel" and c["waves_per_pixel"] == 1
    assert "rccl_probe" not in d and "cpu_baseline" not in d                      # N = 1 extras only
    b = d["step_breakdown_ms"]
    assert b["render"] > 0 and b["all_gather"] > 0
    # every rank's own spans: a slow rank, a slow gather and a slow launch path can be told apart
    assert [r["rank"] for r in b["per_rank"]] == [0, 1]
    for r in b["per_rank"]:
        assert r["render"] > 0 and r["all_gather"] > 0 and r["wall_ms_per_step"] >= r["render"]
        assert abs(r["wall_ms_per_step"] - (r["render"] + r["all_gather"] + r["reassembly"] + r["launch_overhead_ms"])) < 0.01
    for name in ("render", "all_gather", "reassembly", "launch_overhead_ms"):
        sp = b["spread_over_ranks"][name]
        assert sp["min"] <= sp["mean"] <= sp["max"]
    assert b["render"] == b["spread_over_ranks"]["render"]["max"] and "launch_overhead_ms" in b
    # each rank reports the phases it reaches on stderr (a hang is then "rank k stopped after ...")
    for rk in (0, 1):
        for what in ("process group up (gloo)", "ctx created", "warm-up done", "timed 2 steps", "done"):
            assert any(f"rank {rk}/2" in l and what in l for l in p.stderr.splitlines()), (rk, what, p.stderr[-1500:])


def test_bench_abi_multi_mode():
    """`bench.py --abi-multi --gpus 1`: one process, the C ABI's multi-GPU entry (flux_multi_*, RCCL from C) instead of torch.distributed
    ranks -- the mode a node with several GPUs is driven in by a compiled embedder; one JSON line, nothing else on stdout."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--abi-multi", "--gpus", "1", "--config", "3", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 1000.0 and d["config"]["finite"] is True
    assert "flux_multi" in d["config"]["parallelism"] and "ncclAllGather" in d["config"]["parallelism"]
    assert d["multi_info"]["devices"] == 1 and d["multi_info"]["rccl_version"] >= 20000
    b = d["step_breakdown_ms"]
    assert b["render"] > 0 and b["frame_call"] >= b["render"] and b["all_gather"] >= 0
