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
    assert d["rccl_probe"]["ran"] is True and d["rccl_probe"]["backend"] == "nccl" and d["rccl_probe"]["all_gather_equals_local"] is True
