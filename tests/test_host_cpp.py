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
                 "render_manager", "cancel"):
        assert f"ok {name}" in out.stdout


def test_cli_argument_errors(binaries):
    flux = binaries[0]
    r = subprocess.run([flux], capture_output=True, text=True)
    assert r.returncode == 2 and "<scene_file>" in r.stderr
    r = subprocess.run([flux, os.path.join(SCENES, "demo1.yml"), "-n", "host:2000"], capture_output=True, text=True)
    assert r.returncode == 2 and "network" in r.stderr
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
