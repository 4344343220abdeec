"""The N>1 path on CPU: world_size 2, 3 and 8 (the SCALE run's world) over gloo.  Row sharding + one all_gather + reassembly
(flux_amd/dist.py) must reproduce the single-process frame bit for bit.  The pixels come from the CPU
oracle here (no GPU in this container); on GPUs the same FrameSharder is fed by the HIP library."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, small_scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world, *args):
    """mp.spawn on a fresh port; one retry on another port if the rendezvous itself failed (the port found by
    _free_port can be taken by another process between the probe and the bind)."""
    for attempt in range(2):
        port = _free_port()
        try:
            mp.spawn(fn, args=(world, port) + args, nprocs=world, join=True)
            return
        except Exception as ex:  # noqa: BLE001
            msg = str(ex)
            if attempt == 0 and any(k in msg for k in ("address already in use", "Address already in use", "EADDRINUSE",
                                                       "DistNetworkError", "Connection reset", "timed out")):
                continue
            raise


def _worker(rank, world, port, height, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import flux_amd
    from flux_amd.dist import FrameSharder
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = small_scene(flux_amd.load_scene(os.path.join(ROOT, "scenes", "demo2.yml")), 20, height)
    o = oracle.Oracle(sd, flux_amd.JobConfiguration(2, 5, 50), seed=5)

    def render_fn(first, stride, count, out):
        rows = np.arange(first, first + stride * count, stride, dtype=np.int32)
        out[:count] = torch.from_numpy(o.render_row_list(rows))

    sh = FrameSharder(height, 20, rank, world, torch.device("cpu"))
    frame = sh.step(render_fn).clone()
    full = torch.from_numpy(o.render_frame())
    ok = torch.equal(frame, full)
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([int(ok), sh.count]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 16), (2, 15), (3, 16), (8, 19)])   # 8 ranks: the SCALE run's world, ragged (19 rows)
def test_sharded_frame_equals_single(tmp_path, world, height):
    _spawn(_worker, world, height, str(tmp_path))
    counts = []
    for r in range(world):
        ok, cnt = np.load(str(tmp_path / f"ok_{r}.npy"))
        assert ok == 1
        counts.append(int(cnt))
    assert sum(counts) == height and max(counts) - min(counts) <= 1


def test_rank_rows_partition():
    from flux_amd.dist import rank_rows, rows_per_rank
    for h in (1, 7, 600, 601):
        for g in (1, 2, 3, 8):
            rows = []
            for r in range(g):
                first, stride, count = rank_rows(h, r, g)
                rows += list(range(first, first + stride * count, stride))
                assert count <= rows_per_rank(h, g)
            assert sorted(rows) == list(range(h))
    with pytest.raises(ValueError):
        rank_rows(10, 2, 2)


def _set_worker(rank, world, port, width, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import flux_amd
    from flux_amd.dist import SetSharder
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    height = 12
    sd = small_scene(flux_amd.load_scene(os.path.join(ROOT, "scenes", "demo2.yml")), width, height)
    o = oracle.Oracle(sd, flux_amd.JobConfiguration(2, 5, 50), seed=9)
    full = o.render_frame()
    rowperm = np.stack([o.row_perm(r) for r in range(height)])

    def render_fn(first, stride, count, out):  # this rank's pixels: one per row per owned set
        for m in range(count):
            cols = np.argmax(rowperm == first + m * stride, axis=1)
            out[:, m] = torch.from_numpy(full[np.arange(height), cols])

    sh = SetSharder(height, width, rank, world, torch.device("cpu"), torch.from_numpy(rowperm))
    frame = sh.step(render_fn).clone()
    ok = torch.equal(frame, torch.from_numpy(full))
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([int(ok), sh.count]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,width", [(2, 20), (2, 21), (3, 20), (8, 20)])   # 8 ranks: 20 sets as 3,3,3,3,2,2,2,2
def test_set_sharded_frame_equals_single(tmp_path, world, width):
    """SetSharder: ranks own sample sets (s % G == rank), one all_gather, reassembly through the row permutation;
    ragged set counts (21 sets over 2 ranks, 20 over 3) are padded."""
    _spawn(_set_worker, world, width, str(tmp_path))
    counts = []
    for r in range(world):
        ok, cnt = np.load(str(tmp_path / f"ok_{r}.npy"))
        assert ok == 1
        counts.append(int(cnt))
    assert sum(counts) == width and max(counts) - min(counts) <= 1
