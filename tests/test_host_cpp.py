"""C++ host layer (flux_amd/host): the mirror of the reference's worker interface + the flag-compatible
`flux` CLI (flux/src/main.rs:126-205)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SCENES

HOST = os.path.join(ROOT, "flux_amd", "host")


@pytest.fixture(scope="module")
def binaries():
    from flux_amd import build
    build.build_host()
    return os.path.join(HOST, "flux"), os.path.join(HOST, "flux_host_test")


def test_host_self_test(binaries, tmp_path):
    out = subprocess.run([binaries[1], SCENES, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "all ok" in out.stdout
    for name in ("demo1", "demo2", "yaml errors", "work_units", "channel", "image_builder", "bounded channel",
                 "render_manager", "cancel", "cbor", "node messages", "node loopback"):
        assert f"ok {name}" in out.stdout


def test_cli_argument_errors(binaries):
    flux = binaries[0]
    r = subprocess.run([flux], capture_output=True, text=True)
    assert r.returncode == 2 and "<scene_file>" in r.stderr
    # -L -n with nobody listening: "Error connecting to ..." (flux/src/main.rs:60-63); -L alone: no workers
    r = subprocess.run([flux, os.path.join(SCENES, "demo1.yml"), "-L", "-n", "127.0.0.1:1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error connecting to 127.0.0.1:1" in r.stderr
    r = subprocess.run([flux, os.path.join(SCENES, "demo1.yml"), "-L"], capture_output=True, text=True)
    assert r.returncode == 1 and "No workers specified" in r.stderr
    r = subprocess.run([flux, os.path.join(SCENES, "demo1.yml"), "-g"], capture_output=True, text=True)
    assert r.returncode == 2 and "preview" in r.stderr
    r = subprocess.run([flux, os.path.join(SCENES, "demo1.yml"), "-r", "abc"], capture_output=True, text=True)
    assert r.returncode == 2 and "invalid value" in r.stderr
    r = subprocess.run([flux, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 2
    r = subprocess.run([flux, "/nonexistent.yml"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


def test_cli_without_gpu_fails_loudly(binaries, flux):
    if flux._lib.lib.flux_device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([binaries[0], os.path.join(SCENES, "demo1.yml")], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


def _read_ppm(path):
    toks = open(path).read().split()
    assert toks[0] == "P3" and toks[3] == "65535"
    w, h = int(toks[1]), int(toks[2])
    return np.array(toks[4:], dtype=np.int64).reshape(h, w, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [50, 7])
def test_cli_renders_demo1_like_the_oracle(binaries, flux, oracle_mod, demo1, tmp_path, rows):
    """`flux scenes/demo1.yml -r 2 -R rows --seed 1` -> demo1.ppm, compared with the oracle's frame pushed
    through the oracle's Image::write restatement (16-bit quantisation: allow 1 LSB where a value sits on
    a quantisation boundary)."""
    r = subprocess.run([binaries[0], os.path.join(SCENES, "demo1.yml"), "-r", "2", "-d", "5", "-R", str(rows),
                        "--seed", "1", "--gpus", "1", "--outdir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "rendering finished, total time" in r.stdout
    got = _read_ppm(str(tmp_path / "demo1.ppm"))
    o = oracle_mod.Oracle(demo1, flux.JobConfiguration(2, 5, rows), seed=1)
    img = o.render_frame(threads=8)
    units = oracle_mod.work_units(600, rows)
    present = np.zeros(600, dtype=np.uint8)
    for a, b in units:
        present[a:b + 1] = 1
    ref = str(tmp_path / "ref.ppm")
    oracle_mod.write_ppm(ref, img, present)
    want = _read_ppm(ref)
    assert got.shape == want.shape == (600, 800, 3)
    assert np.abs(got - want).max() <= 1
    assert (got != want).mean() < 1e-4
    if rows == 7:  # 600 = 85*7 + 5: last unit is rows 595..599 -> all present; sanity on the partition
        assert present.all()


@pytest.mark.gpu
def test_node_protocol_end_to_end(binaries, tmp_path):
    """`flux_node` (GPU worker behind the reference's CBOR/TCP node protocol) serving `flux -n host:port -L`:
    the frame assembled from the node's RowsReady events equals the frame of a direct local render, bit for
    bit (same seed; colours cross the wire as exact shortest-float CBOR)."""
    import re
    import time
    flux_bin = binaries[0]
    node_bin = os.path.join(HOST, "flux_node")
    direct, remote = tmp_path / "direct", tmp_path / "remote"
    direct.mkdir()
    remote.mkdir()
    scene = os.path.join(SCENES, "demo2.yml")
    common = ["-r", "3", "-d", "5", "-R", "64", "--seed", "5"]
    r = subprocess.run([flux_bin, scene] + common + ["--gpus", "1", "--outdir", str(direct)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    log = open(tmp_path / "node.log", "w")
    node = subprocess.Popen([node_bin, "-h", "127.0.0.1", "-p", "0", "-t", "4", "--seed", "5", "--once"],
                            stdout=log, stderr=subprocess.STDOUT, text=True)
    try:
        port = None
        for _ in range(600):
            m = re.search(r"Listening on port (\d+)", open(tmp_path / "node.log").read())
            if m:
                port = m.group(1)
                break
            assert node.poll() is None, open(tmp_path / "node.log").read()
            time.sleep(0.05)
        assert port, "flux_node did not come up"
        r = subprocess.run([flux_bin, scene] + common + ["-L", "-n", f"127.0.0.1:{port}", "--outdir", str(remote)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "Connecting to worker" in r.stdout and "rendering finished, total time" in r.stdout
        assert node.wait(timeout=30) == 0
    finally:
        if node.poll() is None:
            node.kill()
        log.close()
    a, b = _read_ppm(str(direct / "demo2.ppm")), _read_ppm(str(remote / "demo2.ppm"))
    assert a.shape == (600, 800, 3) and np.array_equal(a, b)
    out = open(tmp_path / "node.log").read()
    assert "Got job" in out and "Got done message" in out


@pytest.mark.gpu
@pytest.mark.parametrize("root,split", [(8, "sets"), (8, "rows"), (3, "sets")])
def test_cli_split_frame_equals_split_units(binaries, tmp_path, root, split):
    """`flux --split sets|rows` (the compiled host's multi-GPU path: ONE MultiGpuWorker over flux_multi_*, i.e. per-device contexts,
    one launch per device, ncclCommInitAll + one ncclAllGather from RCCL's C API, reassembly on the device) writes the same PPM,
    byte for byte, as `--split units` (one GpuWorker per device pulling WorkUnits, the reference's scheme: manager.rs:100,156-162).
    One device here; root 3 is below 64 spp, where `sets` falls back to row tiles.  -R 7: 600 = 85 * 7 + 5 (every row issued)."""
    scene = os.path.join(SCENES, "demo2.yml")
    outs = {}
    for mode in ("units", split):
        d = tmp_path / mode
        d.mkdir()
        r = subprocess.run([binaries[0], scene, "-r", str(root), "-R", "7", "--seed", "9", "--gpus", "1", "--split", mode, "--outdir", str(d)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        if mode != "units":
            assert "one RCCL all-gather" in r.stdout and "multi-GPU frame: create" in r.stdout, r.stdout
        outs[mode] = open(d / "demo2.ppm", "rb").read()
    assert outs[split] == outs["units"]
    r = subprocess.run([binaries[0], scene, "--split", "sets", "-L", "-n", "127.0.0.1:1"], capture_output=True, text=True)
    assert r.returncode == 2 and "--split units" in r.stderr
