"""flux_amd -- MI355X (gfx950) implementation of jtdaugherty/flux's per-pixel render loop.

The package is a thin host layer over flux_amd/libflux_hip.so (hand-written HIP
kernels behind the C ABI of include/flux_abi.h).  Importing it without the
built library raises ImportError: there is no CPU fallback.
"""
from . import _lib
from ._lib import (FluxError, KERNEL_DEFAULT, KERNEL_REFILL, KERNEL_SPLIT, KERNEL_STATIC, MATH_FAST, MATH_STRICT, SHARD_AUTO, SHARD_ROWS,
                   SHARD_SETS)
from .render import MultiRenderer, Renderer, debug_fastmath, release_comms, render_frame_multi, sampler_grid, work_units, write_ppm
from .scene import (CameraData, CameraSettings, EmissiveData, GlossyReflectiveData, JobConfiguration, MatteData,
                    OutputSettings, PlaneData, ReflectiveData, SceneData, SceneError, SphereData, WorkUnit,
                    WorkUnitResult, load_scene, scene_from_dict)

__all__ = [
    "FluxError", "Renderer", "work_units", "write_ppm", "load_scene", "scene_from_dict", "SceneData", "SceneError",
    "CameraSettings", "CameraData", "OutputSettings", "SphereData", "PlaneData", "MatteData", "EmissiveData",
    "ReflectiveData", "GlossyReflectiveData", "JobConfiguration", "WorkUnit", "WorkUnitResult", "KERNEL_DEFAULT",
    "KERNEL_STATIC", "KERNEL_REFILL", "KERNEL_SPLIT", "MATH_FAST", "MATH_STRICT", "debug_fastmath", "sampler_grid",
    "MultiRenderer", "render_frame_multi", "release_comms", "SHARD_AUTO", "SHARD_SETS", "SHARD_ROWS",
]
