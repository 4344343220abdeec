"""Row b' of the scope table from inside a Python process: the multi-GPU frame through the C ABI (flux_multi_*), i.e. the
fan-out of a job to its workers (fluxcore/src/manager.rs:156-162) and ImageBuilder's gather (manager.rs:316-324) as per-device
contexts + ONE ncclAllGather + a reassembly kernel.  This process has torch's RCCL mapped already; the library must use THAT
copy (one RCCL on one HIP runtime).  tests/test_gpu_abi_client.py runs the same entry points from plain C without torch.

A one-GPU box runs RCCL at G = 1 only.  G = 2, 3, 8 -- set shares of unequal size, padding, the reassembly index -- run through
FLUX_SHARD_LOOPBACK (all ranks on device 0, the collective replaced by device-to-device copies; everything else the product path).
"""
import numpy as np
import pytest

from conftest import small_scene

pytestmark = pytest.mark.gpu


def _maps(name):
    with open("/proc/self/maps") as f:
        return sorted({ln.split()[-1] for ln in f if name in ln and ".so" in ln})


@pytest.mark.parametrize("shard", ["sets", "rows"])
def test_g1_rccl_frame_equals_render_rows(flux, demo2, shard):
    sd = small_scene(demo2, 96, 72)
    cfg = flux.JobConfiguration(8, 5, 50)
    with flux.Renderer(sd, cfg, seed=3) as r:
        want = r.render_frame()
    mode = flux.SHARD_SETS if shard == "sets" else flux.SHARD_ROWS
    with flux.MultiRenderer(sd, cfg, seed=3, devices=[0], shard=mode) as m:
        got = m.render_frame()
        again = m.render_frame()
        info, t = m.info(), m.timing()
        ct = m.rank_create_timing(0)
    assert np.array_equal(got, want) and np.array_equal(again, want)
    assert info["devices"] == 1 and info["shard"] == mode and info["rccl_version"] >= 20000
    assert info["share_doubles"] == 72 * 96 * 3
    assert t["kernel_ms"] > 0 and t["all_gather_ms"] >= 0 and t["frame_ms"] >= t["kernel_ms"]
    assert abs(sum(v for k, v in ct.items() if k != "total") - ct["total"]) < 1e-6 * max(ct["total"], 1.0) + 1e-9
    assert len(_maps("librccl")) == 1, _maps("librccl")          # torch's copy, no second RCCL beside it
    assert len(_maps("libamdhip64")) == 1, _maps("libamdhip64")
    # the one-call form, and the cache: the communicator of device list [0] is made once per process
    assert np.array_equal(flux.render_frame_multi(sd, cfg, seed=3, num_devices=1, shard=mode), want)
    with flux.MultiRenderer(sd, cfg, seed=3, devices=[0], shard=mode) as m2:
        assert m2.info()["comms_cached"] == 1


@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("shard", ["sets", "rows"])
def test_loopback_ranks_reassemble_the_frame(flux, demo2, G, shard):
    """W = 50 sets over G = 3 / 8 ranks leaves shares of unequal size (17 + 17 + 16; 7 x 6 + 2 x ... ), H = 37 rows likewise."""
    sd = small_scene(demo2, 50, 37)
    cfg = flux.JobConfiguration(8, 5, 50)
    with flux.Renderer(sd, cfg, seed=11) as r:
        want = r.render_frame()
    mode = (flux.SHARD_SETS if shard == "sets" else flux.SHARD_ROWS) | flux._lib.SHARD_LOOPBACK
    with flux.MultiRenderer(sd, cfg, seed=11, devices=[0] * G, shard=mode) as m:
        got = m.render_frame()
        info = m.info()
    assert np.array_equal(got, want)
    assert info["devices"] == G and info["rccl_version"] == 0 and info["comms_cached"] == 0
    per = -(-50 // G) * 37 if shard == "sets" else -(-37 // G) * 50
    assert info["share_doubles"] == per * 3


def test_auto_picks_rows_below_64_spp_and_rejects_bad_lists(flux, demo1):
    sd = small_scene(demo1, 40, 30)
    cfg = flux.JobConfiguration(4, 5, 50)
    with flux.Renderer(sd, cfg, seed=2) as r:
        want = r.render_frame()
    with flux.MultiRenderer(sd, cfg, seed=2, devices=[0]) as m:
        assert m.info()["shard"] == flux.SHARD_ROWS
        assert np.array_equal(m.render_frame(), want)
    with pytest.raises(flux.FluxError, match="sample_root"):
        flux.MultiRenderer(sd, cfg, seed=2, devices=[0], shard=flux.SHARD_SETS)
    with pytest.raises(flux.FluxError, match="twice"):
        flux.MultiRenderer(sd, cfg, seed=2, devices=[0, 0])
    with pytest.raises(flux.FluxError, match="out of range"):
        flux.MultiRenderer(sd, cfg, seed=2, devices=[0, 99])
    with pytest.raises(flux.FluxError):
        flux.MultiRenderer(sd, cfg, seed=2, devices=[])
