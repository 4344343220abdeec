"""Row (e) on the one GPU a test box has: a world-size-1 NCCL (= RCCL) process group in a FRESH child process
(tests/rccl_child.py).  librccl must load beside libflux_hip.so under the one-HIP-runtime arrangement of flux_amd/_lib.py,
a communicator must come up, and all_gather_into_tensor must take the sharders' f64 [H][cmax][3] / [rows][W][3] device
buffers; the frames assembled from what RCCL returned must equal this (single-process, no process group) frame bit for bit.
Reference: the fan-out and the gather of fluxcore/src/manager.rs:156-162, 316-324.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, small_scene

pytestmark = pytest.mark.gpu

W, H, N, SEED = 64, 48, 8, 7


def test_world1_rccl_group_gathers_the_sharders_buffers(flux, demo2, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), str(tmp_path), str(W), str(H), str(N),
                        str(SEED)], env=env, capture_output=True, text=True, timeout=600)
    sys.stdout.write(p.stdout[-2000:])
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.load(open(tmp_path / "report.json"))
    print("RCCL version", rep["rccl_version"], "backend", rep["backend"], "libs", rep["librccl"])
    assert rep["backend"] == "nccl" and rep["world"] == 1
    assert rep["librccl"], "librccl was not mapped into the child"
    assert len(rep["libflux_hip"]) == 1 and len(rep["libamdhip64"]) == 1, rep  # ONE HIP runtime beside the product library
    assert rep["sets_gather_equals_local"] and rep["rows_gather_equals_local"] and rep["all_reduce_ok"]
    # the single-process frame (this process: no torch.distributed at all), host-buffer entry point
    sd = small_scene(demo2, W, H)
    with flux.Renderer(sd, flux.JobConfiguration(N, 5, 50), seed=SEED) as r:
        ref = r.render_frame()
    assert np.array_equal(np.load(tmp_path / "sets.npy"), ref)
    assert np.array_equal(np.load(tmp_path / "rows.npy"), ref)
