"""Renderer: the Scene + Camera pair of the reference, backed by the HIP library.

    Scene::from_data(job.scene_data, job.config)     fluxcore/src/workers.rs:46
    Camera::new(..., num_sets = image_width, ...)    fluxcore/src/workers.rs:47-54
    camera.render(&scene, unit) -> WorkUnitResult    fluxcore/src/workers.rs:60

All compute happens in libflux_hip.so on the GPU; there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import _lib
from .scene import JobConfiguration, SceneData, SceneDesc, WorkUnit, WorkUnitResult

STAT_NAMES = ("samples", "segments", "matte_bounces", "glossy_bounces", "specular_bounces",
              "emissive_hits", "misses", "depth_exhausted", "bvh_nodes", "tris_tested")


class Renderer:
    def __init__(self, scene_data: SceneData, config: JobConfiguration, seed: int = 1, device: int = 0,
                 set_share=None):
        """`set_share` = (first_set, set_stride): hold the sample tables of this rank's sets only
        (flux_ctx_create_sets; render with render_sets_device)."""
        self.scene_data = scene_data
        self.config = config
        self.seed = int(seed)
        self.device = int(device)
        self.width = scene_data.output_settings.image_width
        self.height = scene_data.output_settings.image_height
        self._desc = SceneDesc(scene_data)
        cfg = _lib.FluxJobCfg(config.sample_root, config.max_trace_depth, config.rows_per_work_unit)
        h = C.c_void_p()
        self.set_share = (0, 1) if set_share is None else (int(set_share[0]), int(set_share[1]))
        _lib.check(_lib.lib.flux_ctx_create_sets(C.byref(self._desc.desc), C.byref(cfg), C.c_uint64(self.seed),
                                                 self.device, self.set_share[0], self.set_share[1], C.byref(h)))
        self._h = h
        self._warn_if_routed()

    def _warn_if_routed(self):
        """FLUX_MATH_FAST is defined for unit surface normals; a scene with a NON-unit plane normal is rendered with the STRICT
        arithmetic (about 3.5 x slower, no traversal kernels for meshes) whatever set_math says -- reported once per renderer
        instead of being found by polling launch_plan()["math"]."""
        if getattr(self, "_routed_warned", False):
            return
        if self.launch_plan(num_rows=1)["route"] == _lib.ROUTE_KEPT_FAST:
            # the other side of the same rule: the scene wants STRICT, the job is too deep for it
            self._routed_warned = True
            import warnings
            warnings.warn("flux_amd: this scene has a plane stored with a non-unit normal, but max_trace_depth leaves no room for the "
                          "STRICT arithmetic's recursion stack (31 levels of LDS, fewer with a mesh): the job stays on FLUX_MATH_FAST "
                          "with long-form glossy weights, and pixels the reference renders as NaN may come out finite. Normalise the "
                          "plane normals or lower max_trace_depth.", RuntimeWarning, stacklevel=3)
            return
        if self.requested_math == _lib.MATH_FAST and self.effective_math() == _lib.MATH_STRICT:
            self._routed_warned = True
            import warnings
            warnings.warn("flux_amd: this scene has a plane stored with a non-unit normal, so FLUX_MATH_FAST renders it with the STRICT "
                          "arithmetic (the reference's operation order, incl. its NaN pixels): several times slower. Normalise the "
                          "plane normals to get the FAST kernels.", RuntimeWarning, stacklevel=3)

    requested_math = _lib.MATH_FAST

    def effective_math(self) -> int:
        """The arithmetic a render really runs with (flux_ctx_launch_plan word 5)."""
        return self.launch_plan(num_rows=1)["math"]

    def __repr__(self):
        names = {_lib.MATH_FAST: "FAST", _lib.MATH_STRICT: "STRICT"}
        try:
            eff = self.effective_math()
            math = names[self.requested_math] + ("" if eff == self.requested_math else f" -> {names[eff]} (non-unit plane normal)")
        except Exception:
            math = "closed"
        return (f"Renderer({self.scene_data.scene_name!r}, {self.width}x{self.height}, sample_root {self.config.sample_root}, "
                f"depth {self.config.max_trace_depth}, seed {self.seed}, device {self.device}, math {math})")

    # -- lifetime ------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            _lib.lib.flux_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if not self._h:
            raise _lib.FluxError(_lib.E_INVALID, "renderer is closed")
        return self._h

    # -- Camera::render --------------------------------------------------------
    def render(self, unit: WorkUnit) -> WorkUnitResult:
        """Camera::render (trace.rs:53-97): rows [row_start,row_end] inclusive."""
        rows = self.render_rows(unit.row_start, unit.row_end)
        return WorkUnitResult(work_unit=unit, rows=rows)

    def render_rows(self, row_start: int, row_end: int) -> np.ndarray:
        if row_start < 0 or row_end < row_start:
            raise _lib.FluxError(_lib.E_INVALID, f"bad row range [{row_start},{row_end}]")
        n = row_end - row_start + 1
        out = np.empty((n, self.width, 3), dtype=np.float64)
        _lib.check(_lib.lib.flux_render_rows(self._handle(), row_start, row_end,
                                             out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def render_rows_device(self, first_row: int, row_stride: int, num_rows: int, d_out_ptr: int, stream: int = 0):
        """Asynchronous device-resident render (see flux_render_rows_device)."""
        _lib.check(_lib.lib.flux_render_rows_device(self._handle(), first_row, row_stride, num_rows,
                                                    C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def render_sets_device(self, first_set: int, set_stride: int, num_sets: int, d_out_ptr: int, stream: int = 0):
        """Asynchronous device-resident render of the pixels whose sample set is first_set + m*set_stride
        (see flux_render_sets_device): out[(row*num_sets + m)*3]."""
        _lib.check(_lib.lib.flux_render_sets_device(self._handle(), first_set, set_stride, num_sets,
                                                    C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def debug_shade(self, origins, directions, depth: int = 1, set_index: int = 0, sample_index: int = 0):
        """Scene::shade on the device for the given rays (flux_debug_shade): (rgb [n,3], first-hit index [n], t [n])."""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        assert o.shape == d.shape
        rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
        n = rays.shape[0]
        rgb = np.empty((n, 3), dtype=np.float64)
        hit = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _lib.check(_lib.lib.flux_debug_shade(self._handle(), n, rays.ctypes.data_as(dp), depth, set_index, sample_index,
                                             rgb.ctypes.data_as(dp), hit.ctypes.data_as(C.POINTER(C.c_int32)),
                                             t.ctypes.data_as(dp)))
        return rgb, hit, t

    def row_perm_table(self) -> np.ndarray:
        """[H][S] int32: the sample-set index of every pixel (row, col)."""
        return np.stack([self.row_perm(r) for r in range(self.height)])

    def render_frame(self) -> np.ndarray:
        return self.render_rows(0, self.height - 1)

    # -- knobs / introspection ---------------------------------------------------
    def set_kernel(self, variant: int):
        _lib.check(_lib.lib.flux_ctx_set_kernel(self._handle(), variant))

    def set_math(self, mode: int):
        """MATH_FAST (default) or MATH_STRICT (reference operation order); include/flux_abi.h."""
        _lib.check(_lib.lib.flux_ctx_set_math(self._handle(), mode))
        self.requested_math = int(mode)
        self._warn_if_routed()

    def last_kernel_ms(self) -> float:
        return float(_lib.lib.flux_ctx_last_kernel_ms(self._handle()))

    def enable_stats(self, on=True):
        _lib.check(_lib.lib.flux_ctx_enable_stats(self._handle(), 1 if on else 0))

    def stats(self, reset=False) -> dict:
        buf = (C.c_uint64 * _lib.NUM_STATS)()
        _lib.check(_lib.lib.flux_ctx_stats(self._handle(), buf, 1 if reset else 0))
        return dict(zip(STAT_NAMES, [int(x) for x in buf]))

    def stats_raw(self) -> list:
        """All FLUX_NUM_STATS slots (slots 10.. are reserved: 0 in the product build; experiment builds with
        -DFLUX_DEBUG_TRIPS count loop trips there, scripts/trip_counts.py)."""
        buf = (C.c_uint64 * _lib.NUM_STATS)()
        _lib.check(_lib.lib.flux_ctx_stats(self._handle(), buf, 0))
        return [int(x) for x in buf]

    def set_traversal(self, mode: int):
        """Extension: 0 = BVH (default), 1 = brute force over the triangles."""
        _lib.check(_lib.lib.flux_ctx_set_traversal(self._handle(), mode))

    def bvh_info(self) -> dict:
        buf = (C.c_uint64 * _lib.BVH_INFO_WORDS)()
        _lib.check(_lib.lib.flux_ctx_bvh_info(self._handle(), buf, _lib.BVH_INFO_WORDS))
        names = ("nodes", "triangles", "max_depth", "max_leaf", "node_bytes", "tri_bytes", "build_us", "wide_nodes",
                 "leaf_records", "fused_leaves", "wide_stack", "wide_node_bytes", "leaf_record_bytes", "wide_in_use")
        return dict(zip(names, [int(x) for x in buf]))

    def launch_plan(self, num_rows=None, num_sets: int = 0) -> dict:
        """What a render call would launch with the current settings (flux_ctx_launch_plan: the library's own launch
        planner): `num_rows` rows of all sets (flux_render_rows*), or `num_sets` of this context's sets over all rows
        (flux_render_sets_device) when num_sets > 0."""
        buf = (C.c_int64 * _lib.PLAN_WORDS)()
        _lib.check(_lib.lib.flux_ctx_launch_plan(self._handle(), self.height if num_rows is None else num_rows, num_sets, buf))
        names = ("kernel", "block", "blocks", "lds", "waves_per_pixel", "math", "route")
        return dict(zip(names, [int(x) for x in buf]))

    def table(self, which: int) -> np.ndarray:
        S = len(range(self.set_share[0], self.width, self.set_share[1]))  # the sets held, in slot order
        N = self.config.sample_root ** 2
        D = self.config.max_trace_depth
        shape = (S, N, 2) if which in (_lib.TABLE_PIXEL, _lib.TABLE_DISC) else (S, D, N, 3)
        out = np.empty(shape, dtype=np.float64)
        _lib.check(_lib.lib.flux_ctx_copy_table(self._handle(), which, out.ctypes.data_as(C.POINTER(C.c_double)),
                                                out.size))
        return out

    def row_perm(self, row: int) -> np.ndarray:
        out = np.empty(self.width, dtype=np.int32)
        _lib.check(_lib.lib.flux_ctx_copy_row_perm(self._handle(), row, out.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   out.size))
        return out

    def camera_basis(self) -> np.ndarray:
        out = np.empty(9, dtype=np.float64)
        _lib.check(_lib.lib.flux_ctx_camera_basis(self._handle(), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.reshape(3, 3)

    def device_bytes(self) -> int:
        return int(_lib.lib.flux_ctx_device_bytes(self._handle()))

    def create_timing(self) -> dict:
        """Where the wall time of this context's creation went, in ms (flux_ctx_create_timing): total, host, runtime,
        alloc, upload, tables, free, other -- the span the reference's timer includes (manager.rs:145 -> 170)."""
        return _create_timing(self._handle())


def _create_timing(handle) -> dict:
    buf = (C.c_double * _lib.CREATE_TIMING_WORDS)()
    _lib.check(_lib.lib.flux_ctx_create_timing(handle, buf))
    return dict(zip(_lib.CREATE_TIMING_NAMES, [float(x) for x in buf]))


class MultiRenderer:
    """One frame on several GPUs of THIS process through the C ABI (flux_multi_*): the job's fan-out to its workers
    (fluxcore/src/manager.rs:156-162) as one context per device, each holding its share of the sample tables, and ImageBuilder's
    gather (manager.rs:316-324) as ONE ncclAllGather (RCCL) + a reassembly kernel on devices[0].  No torch.distributed."""

    def __init__(self, scene_data: SceneData, config: JobConfiguration, seed: int = 1, devices=(0,), shard: int = _lib.SHARD_AUTO):
        self.scene_data, self.config, self.seed = scene_data, config, int(seed)
        self.devices = [int(d) for d in devices]
        self.width = scene_data.output_settings.image_width
        self.height = scene_data.output_settings.image_height
        self._desc = SceneDesc(scene_data)
        cfg = _lib.FluxJobCfg(config.sample_root, config.max_trace_depth, config.rows_per_work_unit)
        devs = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _lib.check(_lib.lib.flux_multi_create(C.byref(self._desc.desc), C.byref(cfg), C.c_uint64(self.seed), devs, len(self.devices),
                                              int(shard), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib.flux_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if not self._h:
            raise _lib.FluxError(_lib.E_INVALID, "multi renderer is closed")
        return self._h

    def render_frame(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float64)
        _lib.check(_lib.lib.flux_multi_render_frame(self._handle(), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def render_frame_device(self) -> int:
        """The frame left on devices[0]: its device address ([H][W][3] f64), valid until the next render call."""
        p = C.c_void_p()
        _lib.check(_lib.lib.flux_multi_render_frame_device(self._handle(), C.byref(p)))
        return int(p.value)

    def set_kernel(self, variant: int):
        _lib.check(_lib.lib.flux_multi_set_kernel(self._handle(), variant))

    def set_math(self, mode: int):
        _lib.check(_lib.lib.flux_multi_set_math(self._handle(), mode))

    def info(self) -> dict:
        buf = (C.c_uint64 * _lib.MULTI_INFO_WORDS)()
        _lib.check(_lib.lib.flux_multi_info(self._handle(), buf))
        names = ("devices", "shard", "rccl_version", "share_doubles", "ctx_bytes", "buffer_bytes", "comms_cached")
        return dict(zip(names, [int(x) for x in buf]))

    def timing(self) -> dict:
        buf = (C.c_double * _lib.MULTI_TIMING_WORDS)()
        _lib.check(_lib.lib.flux_multi_timing(self._handle(), buf))
        names = ("create_ms", "ctx_create_ms", "comm_init_ms", "frame_ms", "kernel_ms", "all_gather_ms", "reassembly_ms", "d2h_ms")
        return dict(zip(names, [float(x) for x in buf]))

    def rank_create_timing(self, rank: int = 0) -> dict:
        h = C.c_void_p()
        _lib.check(_lib.lib.flux_multi_ctx(self._handle(), rank, C.byref(h)))
        return _create_timing(h)

    def rank_renderer(self, rank: int = 0) -> "Renderer":
        """Rank's context as a Renderer (BORROWED: flux_multi_ctx; closing it does nothing) -- for launch_plan, statistics, tables."""
        h = C.c_void_p()
        _lib.check(_lib.lib.flux_multi_ctx(self._handle(), rank, C.byref(h)))
        r = _BorrowedRenderer.__new__(_BorrowedRenderer)
        r.scene_data, r.config, r.seed, r.device = self.scene_data, self.config, self.seed, self.devices[rank]
        r.width, r.height = self.width, self.height
        info = self.info()
        r.set_share = (rank, len(self.devices)) if info["shard"] == _lib.SHARD_SETS else (0, 1)
        r._h = h
        r._routed_warned = True
        return r


class _BorrowedRenderer(Renderer):
    def close(self):
        self._h = None


def render_frame_multi(scene_data: SceneData, config: JobConfiguration, seed: int = 1, num_devices: int = 0,
                       shard: int = _lib.SHARD_AUTO) -> np.ndarray:
    """flux_render_frame_multi: create + render + destroy on the first `num_devices` devices (0 = all)."""
    desc = SceneDesc(scene_data)
    cfg = _lib.FluxJobCfg(config.sample_root, config.max_trace_depth, config.rows_per_work_unit)
    out = np.empty((scene_data.output_settings.image_height, scene_data.output_settings.image_width, 3), dtype=np.float64)
    _lib.check(_lib.lib.flux_render_frame_multi(C.byref(desc.desc), C.byref(cfg), C.c_uint64(int(seed)), None, num_devices, int(shard),
                                                out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def release_comms() -> int:
    return int(_lib.lib.flux_multi_release_comms())


def work_units(image_height: int, rows_per_work_unit: int):
    """Job::work_units (job.rs:65-88) through the C ABI."""
    n = _lib.lib.flux_work_units(image_height, rows_per_work_unit, None, 0)
    _lib.check(n)
    buf = (_lib.FluxWorkUnit * max(n, 1))()
    _lib.check(_lib.lib.flux_work_units(image_height, rows_per_work_unit, buf, n))
    return [WorkUnit(int(buf[i].row_start), int(buf[i].row_end)) for i in range(n)]


def write_ppm(path: str, rgb: np.ndarray, rows_present=None):
    """Image::write (image.rs:43-61) through the C ABI."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w, c = rgb.shape
    assert c == 3
    rp = None
    if rows_present is not None:
        rp_arr = np.ascontiguousarray(rows_present, dtype=np.uint8)
        assert rp_arr.shape == (h,)
        rp = rp_arr.ctypes.data_as(C.POINTER(C.c_uint8))
    _lib.check(_lib.lib.flux_write_ppm(path.encode(), rgb.ctypes.data_as(C.POINTER(C.c_double)), w, h, rp))


def debug_fastmath(fn: int, a, b=None, device: int = 0) -> np.ndarray:
    """out[i] = fn(a[i], b[i]) evaluated on the device by csrc/flux_math.h (test hook)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(a)
    pa = a.ctypes.data_as(C.POINTER(C.c_double))
    pb = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert b.shape == a.shape
        pb = b.ctypes.data_as(C.POINTER(C.c_double))
    _lib.check(_lib.lib.flux_debug_fastmath(device, fn, pa, pb, out.ctypes.data_as(C.POINTER(C.c_double)), a.size))
    return out


def sampler_grid(kind: int, sample_root: int, seed: int = 1, hemi: bool = False, device: int = 0):
    """One set of a samplers-crate generator computed on the device (include/flux_abi.h flux_sampler_grid):
    (n*n, 2) array, plus its to_hemisphere(.., 0.0) image (n*n, 3) when hemi=True."""
    n2 = sample_root * sample_root
    xy = np.empty((n2, 2), dtype=np.float64)
    hm = np.empty((n2, 3), dtype=np.float64) if hemi else None
    _lib.check(_lib.lib.flux_sampler_grid(device, kind, sample_root, seed, xy.ctypes.data_as(C.POINTER(C.c_double)),
                                          hm.ctypes.data_as(C.POINTER(C.c_double)) if hemi else None))
    return (xy, hm) if hemi else xy
