"""Scene description types and the YAML loader.

Mirrors the reference's plain-data types and their serde schema so that the
reference's own scenes/demo*.yml load unchanged:
  SceneData / CameraSettings / CameraData / OutputSettings / ShapeData
      fluxcore/src/scene.rs:12-74
  SphereData / PlaneData / MaterialData and its four payloads
      fluxcore/src/shapes.rs:15-81
  Color as a 3-element sequence                fluxcore/src/color.rs:8-16
  JobConfiguration / WorkUnit                  fluxcore/src/job.rs:40-53
  WorkUnitResult                               fluxcore/src/manager.rs:24-28
serde semantics kept: enums are externally tagged one-key maps
(`Sphere: {...}`, `Matte: {...}`), unknown fields are ignored, missing fields
and unknown variants are errors.
"""
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple, Union

import yaml

from . import _lib

Vec3 = Tuple[float, float, float]


class SceneError(ValueError):
    """Raised where serde_yaml::from_reader would fail (flux/src/main.rs:28-29)."""


def _vec3(v, what) -> Vec3:
    if not isinstance(v, (list, tuple)) or len(v) != 3:
        raise SceneError(f"{what}: expected a sequence of 3 numbers, got {v!r}")
    try:
        return (float(v[0]), float(v[1]), float(v[2]))
    except (TypeError, ValueError):
        raise SceneError(f"{what}: expected a sequence of 3 numbers, got {v!r}")


def _req(m, key, what):
    if not isinstance(m, dict):
        raise SceneError(f"{what}: expected a map, got {m!r}")
    if key not in m:
        raise SceneError(f"{what}: missing field `{key}`")
    return m[key]


def _num(m, key, what) -> float:
    v = _req(m, key, what)
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        raise SceneError(f"{what}.{key}: expected a number, got {v!r}")
    return float(v)


@dataclass
class MatteData:  # shapes.rs:52-56
    diffuse_color: Vec3
    ambient_color: Vec3
    diffuse_coefficient: float


@dataclass
class EmissiveData:  # shapes.rs:61-64
    color: Vec3
    power: float


@dataclass
class ReflectiveData:  # shapes.rs:69-72
    reflect_amount: float
    reflect_color: Vec3


@dataclass
class GlossyReflectiveData:  # shapes.rs:77-81
    reflect_amount: float
    reflect_color: Vec3
    reflect_exponent: float


MaterialData = Union[MatteData, EmissiveData, ReflectiveData, GlossyReflectiveData]  # shapes.rs:42-47


@dataclass
class SphereData:  # shapes.rs:18-23
    center: Vec3
    radius: float
    material: MaterialData
    invert: bool


@dataclass
class PlaneData:  # shapes.rs:33-37
    point: Vec3
    normal: Vec3
    material: MaterialData


ShapeData = Union[SphereData, PlaneData]  # scene.rs:71-74


@dataclass
class CameraSettings:  # scene.rs:14-18
    eye: Vec3
    look_at: Vec3
    up: Vec3


@dataclass
class CameraData:  # scene.rs:53-58
    zoom_factor: float
    view_plane_distance: float
    focal_distance: float
    lens_radius: float


@dataclass
class OutputSettings:  # scene.rs:62-66
    image_width: int
    image_height: int
    pixel_size: float


@dataclass
class SceneData:  # scene.rs:42-49
    scene_name: str
    output_settings: OutputSettings
    background: Vec3
    shapes: List[ShapeData]
    camera_settings: CameraSettings
    camera_data: CameraData


@dataclass
class JobConfiguration:  # job.rs:49-53; defaults of flux/src/main.rs:20-21,172
    sample_root: int = 1
    max_trace_depth: int = 5
    rows_per_work_unit: int = 50


@dataclass
class WorkUnit:  # job.rs:40-44 (row_end inclusive)
    row_start: int
    row_end: int
    job_id: Tuple[int, int] = (0, 0)


@dataclass
class WorkUnitResult:  # manager.rs:24-28; rows: [rows][W][3] float64 ndarray
    work_unit: WorkUnit
    rows: object = field(repr=False, default=None)


def _single_variant(m, what):
    if not isinstance(m, dict) or len(m) != 1:
        raise SceneError(f"{what}: expected an externally tagged enum (one-key map), got {m!r}")
    (tag, body), = m.items()
    return tag, body


def material_from_yaml(m, what="material") -> MaterialData:
    tag, b = _single_variant(m, what)
    w = f"{what}.{tag}"
    if tag == "Matte":
        return MatteData(_vec3(_req(b, "diffuse_color", w), w + ".diffuse_color"),
                         _vec3(_req(b, "ambient_color", w), w + ".ambient_color"),
                         _num(b, "diffuse_coefficient", w))
    if tag == "Emissive":
        return EmissiveData(_vec3(_req(b, "color", w), w + ".color"), _num(b, "power", w))
    if tag == "Reflective":
        return ReflectiveData(_num(b, "reflect_amount", w), _vec3(_req(b, "reflect_color", w), w + ".reflect_color"))
    if tag == "GlossyReflective":
        return GlossyReflectiveData(_num(b, "reflect_amount", w),
                                    _vec3(_req(b, "reflect_color", w), w + ".reflect_color"),
                                    _num(b, "reflect_exponent", w))
    raise SceneError(f"{what}: unknown variant `{tag}`, expected one of "
                     "`Matte`, `Emissive`, `Reflective`, `GlossyReflective`")


def shape_from_yaml(m, what="shape") -> ShapeData:
    tag, b = _single_variant(m, what)
    w = f"{what}.{tag}"
    if tag == "Sphere":
        inv = _req(b, "invert", w)
        if not isinstance(inv, bool):
            raise SceneError(f"{w}.invert: expected a boolean, got {inv!r}")
        return SphereData(_vec3(_req(b, "center", w), w + ".center"), _num(b, "radius", w),
                          material_from_yaml(_req(b, "material", w), w + ".material"), inv)
    if tag == "Plane":
        return PlaneData(_vec3(_req(b, "point", w), w + ".point"), _vec3(_req(b, "normal", w), w + ".normal"),
                         material_from_yaml(_req(b, "material", w), w + ".material"))
    raise SceneError(f"{what}: unknown variant `{tag}`, expected one of `Sphere`, `Plane`")


def _usize(m, key, what) -> int:
    v = _req(m, key, what)
    if isinstance(v, bool) or not isinstance(v, int) or v < 0:
        raise SceneError(f"{what}.{key}: expected an unsigned integer, got {v!r}")
    return v


def scene_from_dict(d) -> SceneData:
    if not isinstance(d, dict):
        raise SceneError("scene: expected a map at top level")
    name = _req(d, "scene_name", "scene")
    if not isinstance(name, str):
        raise SceneError(f"scene.scene_name: expected a string, got {name!r}")
    o = _req(d, "output_settings", "scene")
    cs = _req(d, "camera_settings", "scene")
    cd = _req(d, "camera_data", "scene")
    shapes = _req(d, "shapes", "scene")
    if not isinstance(shapes, list):
        raise SceneError("scene.shapes: expected a sequence")
    return SceneData(
        scene_name=name,
        output_settings=OutputSettings(_usize(o, "image_width", "output_settings"),
                                       _usize(o, "image_height", "output_settings"),
                                       _num(o, "pixel_size", "output_settings")),
        background=_vec3(_req(d, "background", "scene"), "scene.background"),
        shapes=[shape_from_yaml(s, f"shapes[{i}]") for i, s in enumerate(shapes)],
        camera_settings=CameraSettings(_vec3(_req(cs, "eye", "camera_settings"), "camera_settings.eye"),
                                       _vec3(_req(cs, "look_at", "camera_settings"), "camera_settings.look_at"),
                                       _vec3(_req(cs, "up", "camera_settings"), "camera_settings.up")),
        camera_data=CameraData(_num(cd, "zoom_factor", "camera_data"), _num(cd, "view_plane_distance", "camera_data"),
                               _num(cd, "focal_distance", "camera_data"), _num(cd, "lens_radius", "camera_data")),
    )


def load_scene(path) -> SceneData:
    """serde_yaml::from_reader(scene_file) -> SceneData (flux/src/main.rs:27-29)."""
    with open(path, "r") as f:
        try:
            doc = yaml.safe_load(f)
        except yaml.YAMLError as e:
            raise SceneError(f"{path}: {e}")
    return scene_from_dict(doc)


# ---- flattening into the C ABI ----------------------------------------------------

def material_to_abi(m: MaterialData) -> _lib.FluxMaterial:
    out = _lib.FluxMaterial()
    if isinstance(m, MatteData):
        out.kind = _lib.MAT_MATTE
        out.color[:] = m.diffuse_color
        out.ambient[:] = m.ambient_color
        out.k = m.diffuse_coefficient
    elif isinstance(m, EmissiveData):
        out.kind = _lib.MAT_EMISSIVE
        out.color[:] = m.color
        out.k = m.power
    elif isinstance(m, ReflectiveData):
        out.kind = _lib.MAT_REFLECTIVE
        out.color[:] = m.reflect_color
        out.k = m.reflect_amount
    elif isinstance(m, GlossyReflectiveData):
        out.kind = _lib.MAT_GLOSSY
        out.color[:] = m.reflect_color
        out.k = m.reflect_amount
        out.exponent = m.reflect_exponent
    else:
        raise TypeError(f"not a MaterialData: {m!r}")
    return out


class SceneDesc:
    """Owns a flux_scene_desc and the buffers it points to."""

    def __init__(self, sd: SceneData):
        import ctypes as C

        import numpy as np
        analytic, meshes = split_shapes(sd)
        n = len(analytic)
        self.shapes = (_lib.FluxShape * max(n, 1))()
        self.meshes = (_lib.FluxMesh * max(len(meshes), 1))()
        self._mesh_bufs = []
        for i, m in enumerate(meshes):
            v = np.ascontiguousarray(m.vertices, dtype=np.float64).reshape(-1, 3)
            t = np.ascontiguousarray(m.triangles, dtype=np.uint32).reshape(-1, 3)
            self._mesh_bufs.append((v, t))
            fm = self.meshes[i]
            fm.num_vertices = len(v)
            fm.vertices = v.ctypes.data_as(C.POINTER(C.c_double))
            fm.num_triangles = len(t)
            fm.indices = t.ctypes.data_as(C.POINTER(C.c_uint32))
            fm.material = material_to_abi(m.material)
        for i, s in enumerate(analytic):
            fs = self.shapes[i]
            if isinstance(s, SphereData):
                fs.kind = _lib.SHAPE_SPHERE
                fs.p[:] = s.center
                fs.radius = s.radius
                fs.invert = 1 if s.invert else 0
            elif isinstance(s, PlaneData):
                fs.kind = _lib.SHAPE_PLANE
                fs.p[:] = s.point
                fs.n[:] = s.normal
            else:
                raise TypeError(f"not a ShapeData: {s!r}")
            fs.material = material_to_abi(s.material)
        self._name = sd.scene_name.encode()
        d = _lib.FluxSceneDesc()
        d.scene_name = self._name
        d.image_width = sd.output_settings.image_width
        d.image_height = sd.output_settings.image_height
        d.pixel_size = sd.output_settings.pixel_size
        d.background[:] = sd.background
        d.eye[:] = sd.camera_settings.eye
        d.look_at[:] = sd.camera_settings.look_at
        d.up[:] = sd.camera_settings.up
        d.zoom_factor = sd.camera_data.zoom_factor
        d.view_plane_distance = sd.camera_data.view_plane_distance
        d.focal_distance = sd.camera_data.focal_distance
        d.lens_radius = sd.camera_data.lens_radius
        d.num_shapes = n
        d.shapes = self.shapes
        d.num_meshes = len(meshes)
        d.meshes = self.meshes
        self.desc = d


# ---- extension: triangle meshes (absent in the reference, scene.rs:71-74) ------------------------

@dataclass
class MeshData:
    """Indexed triangle mesh.  YAML (extension of the ShapeData enum):
        - Mesh: {vertices: [[x,y,z], ...], triangles: [[i,j,k], ...], material: {...}}
        - Triangle: {v0: [..], v1: [..], v2: [..], material: {...}}      (a one-triangle mesh)
    Hit order: all Sphere/Plane shapes first (YAML order), then mesh triangles in YAML/index order."""
    vertices: object  # float64 ndarray [nv][3]
    triangles: object  # uint32 ndarray [nt][3]
    material: MaterialData


def mesh_from_yaml(tag, b, what):
    import numpy as np
    w = f"{what}.{tag}"
    mat = material_from_yaml(_req(b, "material", w), w + ".material")
    if tag == "Triangle":
        v = np.array([_vec3(_req(b, k, w), f"{w}.{k}") for k in ("v0", "v1", "v2")], dtype=np.float64)
        return MeshData(v, np.array([[0, 1, 2]], dtype=np.uint32), mat)
    verts = _req(b, "vertices", w)
    tris = _req(b, "triangles", w)
    try:
        v = np.array(verts, dtype=np.float64).reshape(-1, 3)
        t = np.array(tris, dtype=np.int64).reshape(-1, 3)
    except (ValueError, TypeError):
        raise SceneError(f"{w}: vertices must be [[x,y,z],...] and triangles [[i,j,k],...]")
    if t.size and (t.min() < 0 or t.max() >= len(v)):
        raise SceneError(f"{w}: triangle index out of range (have {len(v)} vertices)")
    return MeshData(v, t.astype(np.uint32), mat)


_shape_from_yaml_reference = shape_from_yaml


def shape_from_yaml(m, what="shape"):  # noqa: F811  (extends the reference enum with Mesh / Triangle)
    tag, b = _single_variant(m, what)
    if tag in ("Mesh", "Triangle"):
        return mesh_from_yaml(tag, b, what)
    return _shape_from_yaml_reference(m, what)


def split_shapes(sd: SceneData):
    """(analytic shapes in order, meshes in order) -- the hit order of include/flux_abi.h."""
    analytic = [s for s in sd.shapes if not isinstance(s, MeshData)]
    meshes = [s for s in sd.shapes if isinstance(s, MeshData)]
    return analytic, meshes
