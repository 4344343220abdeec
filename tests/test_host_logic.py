"""Host-side pieces of the boundary that need no GPU: Job::work_units and Image::write through the
C ABI, checked against the oracle's restatement."""
import numpy as np
import pytest


@pytest.mark.parametrize("h,r", [(600, 50), (600, 1), (601, 50), (600, 1000), (2, 1), (1, 50), (7, 3), (600, 599)])
def test_work_units_match_oracle(flux, oracle_mod, h, r):
    got = [(u.row_start, u.row_end) for u in flux.work_units(h, r)]
    assert got == oracle_mod.work_units(h, r)


def test_work_units_zero_rows_is_an_error(flux):
    with pytest.raises(flux.FluxError):   # job.rs:67-70 panics
        flux.work_units(600, 0)


def test_work_units_cover_rows_in_order(flux):
    us = flux.work_units(600, 50)
    assert len(us) == 12
    rows = [r for u in us for r in range(u.row_start, u.row_end + 1)]
    assert rows == list(range(600))
    assert len(flux.work_units(600, 1)) == 599  # the reference never issues the last single row


def test_write_ppm_matches_oracle(flux, oracle_mod, tmp_path):
    rng = np.random.default_rng(3)
    img = rng.random((9, 7, 3)) * 1.2 - 0.1   # includes <0 and >1 values
    img[0, 0] = [1.0, 0.5, 0.0]
    img[1, 1] = [np.nan, np.inf, -np.inf]
    present = np.array([1, 1, 0, 1, 1, 1, 0, 1, 1], dtype=np.uint8)
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    flux.write_ppm(a, img, present)
    oracle_mod.write_ppm(b, img, present)
    ta, tb = open(a).read(), open(b).read()
    assert ta == tb
    lines = ta.splitlines()
    assert lines[:4] == ["P3", "7 9", "65535", "65535 32767 0"] and len(lines) == 3 + 63
    flux.write_ppm(a, img)
    oracle_mod.write_ppm(b, img)
    assert open(a).read() == open(b).read()


def test_write_ppm_io_error(flux):
    with pytest.raises(flux.FluxError):
        flux.write_ppm("/nonexistent_dir/x.ppm", np.zeros((1, 1, 3)))


def test_no_gpu_means_loud_failure(flux, demo1):
    """On a box without a GPU the product refuses to render (no CPU fallback)."""
    if flux._lib.lib.flux_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(flux.FluxError, match="no HIP device"):
        flux.Renderer(demo1, flux.JobConfiguration(1, 5, 50))


def test_bench_self_launches_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus N` from a bare shell (what the driver runs) starts N ranks itself -- fresh interpreters
    under torch.distributed.run, spawned before anything touches a GPU -- and the whole launch exits non-zero when a rank
    fails.  Here (no GPU) every rank refuses to run: there is no CPU fallback to report a number from."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    import flux_amd
    if flux_amd._lib.lib.flux_device_count() >= 2:
        assert p.returncode == 0 and '"n_gpus": 2' in p.stdout, p.stderr[-2000:]
    else:
        assert p.returncode != 0
        assert p.stdout.strip() == ""                                  # no bench line without the GPUs
        # (the launcher ends the other rank as soon as the first one has failed: under load that rank may not have got as far as its own
        # refusal, or even its imports -- one refusal and one "imported" line are what every run shows)
        assert p.stderr.count("needs a GPU") >= 1 or "invalid device ordinal" in p.stderr or "out of range" in p.stderr
        # ... and the launcher says how far every rank got (round 5: a failed or hung 8-GPU run must explain itself)
        assert "last phase each rank reached" in p.stderr
        lines = [l.strip() for l in p.stderr.splitlines()]
        for rk in (0, 1):
            assert any(l.startswith(f"rank {rk}:") for l in lines), p.stderr[-1500:]
        assert any(l.startswith("rank ") and "imported torch + flux_amd" in l for l in lines), p.stderr[-1500:]


def test_bench_launch_is_bounded_and_says_where_it_stopped():
    """A multi-rank bench that does not finish within FLUX_BENCH_LAUNCH_TIMEOUT_S is ended by its launcher -- the launcher's own
    process group, by id, never a pattern -- with a non-zero exit and one line per rank naming the last phase it reached
    (a first 8-GPU run must explain its own failure).  Here the limit (2 s) ends the ranks while they are
    still importing torch."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, FLUX_BENCH_LAUNCH_TIMEOUT_S="2")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert p.returncode == 124, (p.returncode, p.stderr[-1000:])
    assert p.stdout.strip() == ""
    assert "exceeded 2.0 s" in p.stderr and "rank 0:" in p.stderr and "rank 1:" in p.stderr
