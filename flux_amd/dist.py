"""Multi-GPU frame rendering: image rows shard across ranks, one framebuffer gather.

In the reference the same two steps are the shared work-unit channel handed to every worker
(fluxcore/src/manager.rs:100,156-162) and ImageBuilder placing RowsReady rows by row_start
(manager.rs:316-324).  Here the partition is static -- rank r renders rows r, r+G, r+2G, ... so cheap
sky rows and expensive sphere rows are spread evenly -- and the reassembly is ONE collective
(all_gather over RCCL/xGMI on GPUs, gloo on CPU for tests) followed by a stride-interleave view.
Pixels are independent given (seed, row), so the assembled image is bit-identical for every G.

torch is plumbing here (device buffers, streams, the process group); the pixels come from the
`render_fn` callable (Renderer.render_rows_device via hip_render_fn below; the gloo tests inject
their own CPU checker).
"""
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def _all_gather(gathered: torch.Tensor, local: torch.Tensor, group=None):
    """all_gather_into_tensor; over gloo (tests, rehearsals: several ranks on one GPU) device tensors go through the host."""
    if local.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.all_gather_into_tensor(host.view(-1), local.detach().cpu().view(-1), group=group)
        gathered.copy_(host)
    else:
        dist.all_gather_into_tensor(gathered.view(-1), local.view(-1), group=group)


def rank_rows(height: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(first_row, row_stride, num_rows) of `rank`'s share of an image of `height` rows."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    count = (height - rank + world - 1) // world if height > rank else 0
    return rank, world, count


def rows_per_rank(height: int, world: int) -> int:
    """Padded per-rank row count (rank 0 always has the most rows)."""
    return (height + world - 1) // world


class FrameSharder:
    """Owns the per-rank row buffer and the gathered frame for one image geometry."""

    def __init__(self, height: int, width: int, rank: int, world: int, device, group=None):
        self.height, self.width, self.rank, self.world = height, width, rank, world
        self.group = group
        self.first, self.stride, self.count = rank_rows(height, rank, world)
        self.rmax = rows_per_rank(height, world)
        # zero-initialised: padded rows (ranks with one row fewer) stay zero, like Image::write's
        # zero padding of never-received rows (image.rs:55-59)
        self.local = torch.zeros((self.rmax, width, 3), dtype=torch.float64, device=device)
        self.gathered = torch.zeros((world, self.rmax, width, 3), dtype=torch.float64, device=device)

    def render(self, render_fn: Callable[[int, int, int, torch.Tensor], None]):
        """render_fn(first_row, row_stride, num_rows, out) fills out[:num_rows] ([rows][W][3] f64)."""
        if self.count:
            render_fn(self.first, self.stride, self.count, self.local)

    def collect(self):
        """The ONE collective of a frame: all_gather of every rank's row buffer (RCCL over xGMI on GPUs)."""
        if self.world > 1:
            _all_gather(self.gathered, self.local, self.group)

    def assemble(self) -> torch.Tensor:
        """The gathered buffers in image order: [H][W][3] (a view, valid on every rank)."""
        if self.world == 1:
            return self.local[: self.height]
        # gathered[g][k] is image row k*G + g  ->  [k][g] order is image order
        return self.gathered.permute(1, 0, 2, 3).reshape(self.rmax * self.world, self.width, 3)[: self.height]

    def gather(self) -> torch.Tensor:
        self.collect()
        return self.assemble()

    def step(self, render_fn) -> torch.Tensor:
        self.render(render_fn)
        return self.gather()


class SetSharder:
    """Shards the frame by SAMPLE SET instead of by row: rank g renders, in every row, the pixels whose set
    index s satisfies s % G == g (flux_render_sets_device).  Every row's pixels use a permutation of all sets
    (trace.rs:64-69), so each rank gets exactly one pixel per row per owned set: a 1/G share with the cost of
    an average pixel (perfect balance), and -- unlike row shards -- each rank keeps the table locality of the
    full-frame render (600 rows per set, one set per XCD at a time).  Reassembly is still ONE all_gather, then
    one indexed read with the row permutation.  `rowperm` is the [H][W] set index of every pixel."""

    def __init__(self, height: int, width: int, rank: int, world: int, device, rowperm: torch.Tensor, group=None):
        self.height, self.width, self.rank, self.world = height, width, rank, world
        self.group = group
        num_sets = width  # workers.rs:50: num_sets = image_width
        self.count = len(range(rank, num_sets, world))
        self.cmax = (num_sets + world - 1) // world
        self.render_buf = torch.zeros((height, self.count, 3), dtype=torch.float64, device=device)
        padded = self.count != self.cmax
        self.local = torch.zeros((height, self.cmax, 3), dtype=torch.float64, device=device) if padded else self.render_buf
        self.gathered = torch.zeros((world, height, self.cmax, 3), dtype=torch.float64, device=device)
        rp = rowperm.to(device=device, dtype=torch.int64)
        assert rp.shape == (height, width)
        self._g = rp % world       # which rank rendered pixel (r, c)
        self._m = rp // world      # its position among that rank's sets
        self._r = torch.arange(height, device=device, dtype=torch.int64).unsqueeze(1).expand(height, width)

    def render(self, render_fn: Callable[[int, int, int, torch.Tensor], None]):
        """render_fn(first_set, set_stride, num_sets, out) fills out[H][num_sets][3] (f64)."""
        if self.count:
            render_fn(self.rank, self.world, self.count, self.render_buf)

    def collect(self):
        """The ONE collective of a frame: all_gather of every rank's [H][sets][3] share."""
        if self.local is not self.render_buf:
            self.local[:, : self.count] = self.render_buf
        if self.world > 1:
            _all_gather(self.gathered, self.local, self.group)

    def assemble(self) -> torch.Tensor:
        """One indexed read with the row permutation: pixel (r, c) uses set s = rowperm[r][c], rendered by rank s % G
        as its (s // G)-th set -- the counterpart of ImageBuilder placing rows (manager.rs:316-324)."""
        src = self.local.unsqueeze(0) if self.world == 1 else self.gathered
        return src[self._g, self._r, self._m]  # [H][W][3]

    def gather(self) -> torch.Tensor:
        self.collect()
        return self.assemble()

    def step(self, render_fn) -> torch.Tensor:
        self.render(render_fn)
        return self.gather()


def hip_render_sets_fn(renderer):
    """SetSharder render_fn backed by the HIP library, launched on torch's current stream."""

    def fn(first, stride, count, out):
        stream = torch.cuda.current_stream(out.device).cuda_stream
        renderer.render_sets_device(first, stride, count, out.data_ptr(), stream)

    return fn


def hip_render_fn(renderer):
    """render_fn backed by the HIP library, launched on torch's current stream."""

    def fn(first, stride, count, out):
        stream = torch.cuda.current_stream(out.device).cuda_stream
        renderer.render_rows_device(first, stride, count, out.data_ptr(), stream)

    return fn
