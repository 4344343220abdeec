"""ctypes binding of the C ABI in include/flux_abi.h (flux_amd/libflux_hip.so).

There is no fallback: if the library is missing or a symbol is absent, importing
this module raises.  Compute entry points fail with FluxError when no HIP
device is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLUX_HIP_LIB: experiment builds of the same sources (scripts/sweep_variants.py); never a fallback.
LIB_PATH = os.environ.get("FLUX_HIP_LIB") or os.path.join(_HERE, "libflux_hip.so")

FLUX_OK = 0
E_INVALID, E_DEVICE, E_NOMEM, E_IO = -1, -2, -3, -4
SHAPE_SPHERE, SHAPE_PLANE = 0, 1
MAT_MATTE, MAT_EMISSIVE, MAT_REFLECTIVE, MAT_GLOSSY = 0, 1, 2, 3
KERNEL_DEFAULT, KERNEL_STATIC, KERNEL_REFILL, KERNEL_SPLIT = 0, 1, 2, 3
MATH_FAST, MATH_STRICT = 0, 1
SAMPLER_REGULAR, SAMPLER_JITTERED, SAMPLER_MULTI_JITTERED, SAMPLER_CORRELATED_MULTI_JITTERED = 0, 1, 2, 3
TABLE_PIXEL, TABLE_DISC, TABLE_HEMI = 0, 1, 2
TRAVERSE_BVH, TRAVERSE_BRUTE, TRAVERSE_BVH_BINARY = 0, 1, 2
NUM_STATS = 16
BVH_INFO_WORDS = 16
PLAN_WORDS = 8
PLAN_NONE, PLAN_STATIC, PLAN_REFILL, PLAN_SPLIT, PLAN_BVH_BINARY, PLAN_BVH4 = -1, 0, 1, 2, 3, 4
ROUTE_NONE, ROUTE_TO_STRICT, ROUTE_KEPT_FAST = 0, 1, 2
CREATE_TIMING_WORDS = 8
CREATE_TIMING_NAMES = ("total", "host", "runtime", "alloc", "upload", "tables", "free", "other")
SHARD_AUTO, SHARD_SETS, SHARD_ROWS = 0, 1, 2
SHARD_LOOPBACK = 0x100
MULTI_INFO_WORDS = 8
MULTI_TIMING_WORDS = 8
ABI_VERSION = 3


class FluxMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("color", C.c_double * 3),
                ("ambient", C.c_double * 3), ("k", C.c_double), ("exponent", C.c_double)]


class FluxShape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("invert", C.c_int32), ("p", C.c_double * 3), ("n", C.c_double * 3),
                ("radius", C.c_double), ("material", FluxMaterial)]


class FluxMesh(C.Structure):
    _fields_ = [("num_vertices", C.c_uint64), ("vertices", C.POINTER(C.c_double)),
                ("num_triangles", C.c_uint64), ("indices", C.POINTER(C.c_uint32)), ("material", FluxMaterial)]


class FluxSceneDesc(C.Structure):
    _fields_ = [("scene_name", C.c_char_p), ("image_width", C.c_uint64), ("image_height", C.c_uint64),
                ("pixel_size", C.c_double), ("background", C.c_double * 3), ("eye", C.c_double * 3),
                ("look_at", C.c_double * 3), ("up", C.c_double * 3), ("zoom_factor", C.c_double),
                ("view_plane_distance", C.c_double), ("focal_distance", C.c_double),
                ("lens_radius", C.c_double), ("num_shapes", C.c_uint64), ("shapes", C.POINTER(FluxShape)),
                ("num_meshes", C.c_uint64), ("meshes", C.POINTER(FluxMesh))]


class FluxJobCfg(C.Structure):
    _fields_ = [("sample_root", C.c_uint64), ("max_trace_depth", C.c_uint64),
                ("rows_per_work_unit", C.c_uint64)]


class FluxWorkUnit(C.Structure):
    _fields_ = [("row_start", C.c_uint64), ("row_end", C.c_uint64)]


class FluxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"flux error {code}: {msg}")
        self.code = code


# every symbol include/flux_abi.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "flux_abi_version": (C.c_uint32, []),
    "flux_last_error": (C.c_char_p, []),
    "flux_device_count": (C.c_int, []),
    "flux_build_id": (C.c_char_p, []),
    "flux_device_warmup": (C.c_int, [C.c_int]),
    "flux_ctx_create": (C.c_int, [C.POINTER(FluxSceneDesc), C.POINTER(FluxJobCfg), C.c_uint64, C.c_int,
                                  C.POINTER(_P)]),
    "flux_ctx_create_sets": (C.c_int, [C.POINTER(FluxSceneDesc), C.POINTER(FluxJobCfg), C.c_uint64, C.c_int, C.c_uint64,
                                       C.c_uint64, C.POINTER(_P)]),
    "flux_ctx_destroy": (None, [_P]),
    "flux_render_rows": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]),
    "flux_render_rows_device": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P]),
    "flux_render_sets_device": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P]),
    "flux_ctx_set_kernel": (C.c_int, [_P, C.c_int]),
    "flux_ctx_set_traversal": (C.c_int, [_P, C.c_int]),
    "flux_ctx_set_math": (C.c_int, [_P, C.c_int]),
    "flux_debug_shade": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_double), C.c_uint64, C.c_uint64, C.c_uint64,
                                   C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "flux_sampler_grid": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]),
    "flux_debug_fastmath": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.c_uint64]),
    "flux_ctx_bvh_info": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_uint64]),
    "flux_ctx_launch_plan": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.POINTER(C.c_int64)]),
    "flux_ctx_last_kernel_ms": (C.c_double, [_P]),
    "flux_ctx_enable_stats": (C.c_int, [_P, C.c_int]),
    "flux_ctx_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int]),
    "flux_ctx_copy_table": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.c_uint64]),
    "flux_ctx_copy_row_perm": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_int32), C.c_uint64]),
    "flux_ctx_camera_basis": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "flux_ctx_device_bytes": (C.c_uint64, [_P]),
    "flux_ctx_create_timing": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "flux_multi_create": (C.c_int, [C.POINTER(FluxSceneDesc), C.POINTER(FluxJobCfg), C.c_uint64, C.POINTER(C.c_int), C.c_uint64,
                                    C.c_int, C.POINTER(_P)]),
    "flux_multi_destroy": (None, [_P]),
    "flux_multi_render_frame": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "flux_multi_render_frame_device": (C.c_int, [_P, C.POINTER(_P)]),
    "flux_multi_set_kernel": (C.c_int, [_P, C.c_int]),
    "flux_multi_set_math": (C.c_int, [_P, C.c_int]),
    "flux_multi_ctx": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "flux_multi_info": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "flux_multi_timing": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "flux_multi_release_comms": (C.c_int, []),
    "flux_render_frame_multi": (C.c_int, [C.POINTER(FluxSceneDesc), C.POINTER(FluxJobCfg), C.c_uint64, C.POINTER(C.c_int),
                                          C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "flux_work_units": (C.c_int64, [C.c_uint64, C.c_uint64, C.POINTER(FluxWorkUnit), C.c_uint64]),
    "flux_write_ppm": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.c_uint64, C.c_uint64,
                                 C.POINTER(C.c_uint8)]),
}


def _one_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel ships its own libamdhip64.so (SONAME libamdhip64.so.7, the
    name libflux_hip.so links against).  Loaded first, it satisfies this library's dependency and both share one
    runtime -- streams and device buffers handed over by flux_amd/dist.py are then valid on both sides.  Loaded
    second, it comes in BESIDE /opt/rocm's copy and `torch.cuda` can no longer initialise the device.  So when torch is
    installed it is imported before the library is loaded (FLUX_NO_TORCH=1 skips this for torch-free embedders)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("FLUX_NO_TORCH"):
        return
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def _load():
    _one_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python flux_amd/build.py` "
            "(the renderer has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    # the version FIRST: a stale library lacks newer symbols, and the loop below would die on the first of them with a bare
    # AttributeError instead of saying what to do
    try:
        lib.flux_abi_version.restype = C.c_uint32
        lib.flux_abi_version.argtypes = []
        have = int(lib.flux_abi_version())
    except AttributeError:
        have = 1  # version 1 had no such symbol
    if have != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} speaks ABI version {have}, this binding {ABI_VERSION}: rebuild it "
                          "(python flux_amd/build.py --force)")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error():
    return (lib.flux_last_error() or b"").decode("utf-8", "replace")


def check(rc):
    if rc < 0:
        raise FluxError(rc, last_error())
    return rc
