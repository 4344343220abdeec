"""Procedural triangle scenes (extension: the reference has no triangles, so it defines no such scene;
BASELINE.json config 5 asks for a "procedural 1M-triangle scene (stress BVH)").

heightfield_scene() is fully specified here so that the CPU checker, the GPU path and any future reader
build the identical mesh:

  * demo2.yml's camera, output settings, background, environment sphere (shape 0) and area light
    (shape 1) and its ten glossy spheres (shapes 2..11) are kept; demo2's ground Plane is replaced by
  * an infinite Matte plane at y = -1 (catches rays that leave the height field), and
  * ONE Matte (0.5,0.5,0.5; kd 1) mesh: a regular (nx+1) x (nz+1) vertex grid over
    x in [-14, 14], z in [-10, 20]; vertex (i,j) at x = -14 + 28 i/nx, z = -10 + 30 j/nz,
    y = 0.35 sin(1.7 x) cos(1.3 z) + 0.02 (2 u_ij - 1), u_ij = numpy default_rng(seed).random((nx+1, nz+1))[i, j];
    cell (i,j) gives triangles (v00, v01, v11) and (v00, v11, v10) with v_ab = vertex (i+a, j+b),
    cells in i-major order -> 2 nx nz triangles (nx=1000, nz=500: exactly 1,000,000).
"""
import copy
import os

import numpy as np

from .scene import MatteData, MeshData, PlaneData, SceneData, load_scene

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def heightfield_mesh(nx: int, nz: int, seed: int = 12345) -> MeshData:
    i = np.arange(nx + 1, dtype=np.float64)
    j = np.arange(nz + 1, dtype=np.float64)
    x = -14.0 + 28.0 * i / nx
    z = -10.0 + 30.0 * j / nz
    X, Z = np.meshgrid(x, z, indexing="ij")
    u = np.random.default_rng(seed).random((nx + 1, nz + 1))
    Y = 0.35 * np.sin(1.7 * X) * np.cos(1.3 * Z) + 0.02 * (2.0 * u - 1.0)
    verts = np.stack([X, Y, Z], axis=-1).reshape(-1, 3)

    def vid(a, b):
        return (a * (nz + 1) + b).astype(np.uint32)

    I, J = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
    I, J = I.reshape(-1), J.reshape(-1)
    v00, v01, v11, v10 = vid(I, J), vid(I, J + 1), vid(I + 1, J + 1), vid(I + 1, J)
    tris = np.empty((nx * nz, 2, 3), dtype=np.uint32)
    tris[:, 0, 0], tris[:, 0, 1], tris[:, 0, 2] = v00, v01, v11
    tris[:, 1, 0], tris[:, 1, 1], tris[:, 1, 2] = v00, v11, v10
    return MeshData(verts, tris.reshape(-1, 3), MatteData((0.5, 0.5, 0.5), (1.0, 1.0, 1.0), 1.0))


def heightfield_scene(nx: int = 1000, nz: int = 500, seed: int = 12345, base: SceneData = None) -> SceneData:
    sd = copy.deepcopy(base) if base is not None else load_scene(os.path.join(_ROOT, "scenes", "demo2.yml"))
    sd.scene_name = f"heightfield_{nx}x{nz}"
    shapes = [s for s in sd.shapes if not isinstance(s, PlaneData)]
    shapes.append(PlaneData((0.0, -1.0, 0.0), (0.0, 1.0, 0.0), MatteData((0.5, 0.5, 0.5), (1.0, 1.0, 1.0), 1.0)))
    shapes.append(heightfield_mesh(nx, nz, seed))
    sd.shapes = shapes
    return sd
