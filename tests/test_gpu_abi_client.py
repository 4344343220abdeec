"""A plain-C client of include/flux_abi.h (tests/abi_c_client.c: no Python, no C++ host layer) built with gcc, run as
its own process and compared bit for bit with the ctypes path -- the boundary behaves the same for any embedder
(the reference's would be Rust: INTEGRATION.md)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, small_scene

pytestmark = pytest.mark.gpu


def build_client(tmp_path):
    exe = str(tmp_path / "abi_c_client")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_c_client.c"), "-o", exe,
                    "-L" + os.path.join(ROOT, "flux_amd"), "-lflux_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "flux_amd"), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


@pytest.mark.parametrize("width,height,root,rows", [(64, 48, 4, 50), (80, 60, 8, 7)])
def test_c_client_equals_python_binding(flux, demo1, tmp_path, width, height, root, rows):
    exe = build_client(tmp_path)
    out = str(tmp_path / "frame.bin")
    p = subprocess.run([exe, out, str(width), str(height), str(root), "7", str(rows)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, dtype=np.float64).reshape(height, width, 3)
    sd = small_scene(demo1, width, height)
    with flux.Renderer(sd, flux.JobConfiguration(root, 5, rows), seed=7) as r:
        want = r.render_frame()
    units = flux.work_units(height, rows)
    covered = np.zeros(height, dtype=bool)
    for u in units:
        covered[u.row_start:u.row_end + 1] = True
    assert np.array_equal(got[covered], want[covered])           # bit for bit
    assert not got[~covered].any()                               # job.rs:74's last-row quirk: never issued, stays zero


@pytest.mark.parametrize("width,height,root,shard", [(80, 60, 8, 1), (80, 60, 8, 2), (64, 48, 4, 0)])
def test_c_client_multi_gpu_entry_equals_render_rows(flux, demo1, tmp_path, width, height, root, shard):
    """Row b' of the scope table: the multi-GPU frame through the C ABI alone (flux_multi_create / flux_multi_render_frame /
    flux_render_frame_multi; fan-out + gather of fluxcore/src/manager.rs:156-162, 316-324), at G = 1 -- all a one-GPU box can
    run: per-device context holding its set share, one launch, ncclCommInitAll + ONE ncclAllGather from RCCL's C API (no torch in
    that process), reassembly by the row permutation on the device.  The client itself compares the frames with memcmp; here the
    file it wrote is compared with the Python binding's frame as well.  shard 1 = sample sets, 2 = interleaved rows, 0 = auto (rows
    below 64 spp)."""
    exe = build_client(tmp_path)
    out = str(tmp_path / "frame.bin")
    p = subprocess.run([exe, out, str(width), str(height), str(root), "7", "50", "multi", "1", str(shard)], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr + p.stdout
    info = dict(zip(*[iter(p.stdout.split("multi: ")[1].splitlines()[0].split())] * 2))
    assert int(info["devices"]) == 1 and int(info["rccl"]) >= 20000 and int(info["cached"]) == 0
    assert int(info["shard"]) == (shard if shard else (1 if root * root >= 64 else 2))
    got = np.fromfile(out, dtype=np.float64).reshape(height, width, 3)
    sd = small_scene(demo1, width, height)
    with flux.Renderer(sd, flux.JobConfiguration(root, 5, 50), seed=7) as r:
        want = r.render_frame()
    assert np.array_equal(got, want)
