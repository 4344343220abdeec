"""ctypes binding of the CPU oracle (oracle/libflux_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under flux_amd/.
Pinned to the reference's demo.png at 16 bits (statistical) + hand-derived KATs: see oracle/flux_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflux_oracle.so")

SHAPE_SPHERE, SHAPE_PLANE = 0, 1
MAT_MATTE, MAT_EMISSIVE, MAT_REFLECTIVE, MAT_GLOSSY = 0, 1, 2, 3
KIND_PIXEL, KIND_DISC, KIND_HEMI, KIND_ROWPERM = 1, 2, 3, 4
STAT_NAMES = ("samples", "segments", "matte_bounces", "glossy_bounces", "specular_bounces",
              "emissive_hits", "misses", "depth_exhausted")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _build_if_needed():
    src = [os.path.join(_HERE, f) for f in ("flux_oracle.c", "flux_oracle.h")]
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in src):
        return
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def _load():
    _build_if_needed()
    lib = C.CDLL(LIB_PATH)
    lib.fxo_ctx_create.restype = C.c_void_p
    lib.fxo_ctx_create.argtypes = [_dp, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _ip, _dp, _ip, _dp, C.c_int,
                                   C.c_int, C.c_uint64]
    lib.fxo_ctx_destroy.argtypes = [C.c_void_p]
    lib.fxo_ctx_add_mesh.argtypes = [C.c_void_p, _dp, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, C.c_int, _dp]
    lib.fxo_render_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, _dp, C.c_int]
    lib.fxo_render_row_list.argtypes = [C.c_void_p, _ip, C.c_size_t, _dp, C.c_int]
    lib.fxo_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.fxo_stats_reset.argtypes = [C.c_void_p]
    for f in ("fxo_pixel_sets", "fxo_disc_sets", "fxo_hemi_sets"):
        getattr(lib, f).restype = _dp
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.fxo_row_perm.argtypes = [C.c_void_p, C.c_size_t, _ip]
    lib.fxo_camera_basis.argtypes = [C.c_void_p, _dp]
    lib.fxo_rng_draw.restype = C.c_uint64
    lib.fxo_rng_draw.argtypes = [C.c_uint64, C.c_uint64]
    lib.fxo_rng_key.restype = C.c_uint64
    lib.fxo_rng_key.argtypes = [C.c_uint64] * 5
    lib.fxo_rng_unit.restype = C.c_double
    lib.fxo_rng_unit.argtypes = [C.c_uint64, C.c_uint64]
    lib.fxo_shuffle.argtypes = [C.c_uint64, _ip, C.c_size_t]
    lib.fxo_grid_regular.argtypes = [C.c_int, _dp]
    lib.fxo_grid_jittered.argtypes = [C.c_uint64, C.c_int, _dp]
    lib.fxo_grid_multi_jittered.argtypes = [C.c_uint64] * 4 + [C.c_int, _dp]
    lib.fxo_grid_correlated_multi_jittered.argtypes = [C.c_uint64] * 4 + [C.c_int, _dp]
    lib.fxo_to_unit_hemi.argtypes = [C.c_double, C.c_double, C.c_double, _dp]
    lib.fxo_to_poisson_disc.argtypes = [C.c_double, C.c_double, _dp]
    lib.fxo_max_to_one.argtypes = [_dp]
    lib.fxo_bbox_hit.argtypes = [_dp] * 4
    lib.fxo_sphere_hit.argtypes = [_dp, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp]
    lib.fxo_plane_hit.argtypes = [_dp] * 7
    lib.fxo_scene_hit.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
    lib.fxo_shade.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_size_t, C.c_size_t, _dp]
    lib.fxo_primary_ray.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _dp, _dp]
    lib.fxo_sample_f.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.fxo_work_units.restype = C.c_size_t
    lib.fxo_work_units.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t]
    lib.fxo_ppm_quantize.restype = C.c_uint16
    lib.fxo_ppm_quantize.argtypes = [C.c_double]
    lib.fxo_write_ppm.argtypes = [C.c_char_p, _dp, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint8)]
    return lib


lib = _load()


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def vec(*xs):
    return (C.c_double * len(xs))(*xs)


def material_params(m):
    """(kind, 8 doubles) from a flux_amd.scene material dataclass (duck-typed by field names)."""
    p = [0.0] * 8
    n = type(m).__name__
    if n == "MatteData":
        p[0:3] = m.diffuse_color
        p[3:6] = m.ambient_color
        p[6] = m.diffuse_coefficient
        return MAT_MATTE, p
    if n == "EmissiveData":
        p[0:3] = m.color
        p[3] = m.power
        return MAT_EMISSIVE, p
    if n == "ReflectiveData":
        p[0:3] = m.reflect_color
        p[3] = m.reflect_amount
        return MAT_REFLECTIVE, p
    if n == "GlossyReflectiveData":
        p[0:3] = m.reflect_color
        p[3] = m.reflect_amount
        p[4] = m.reflect_exponent
        return MAT_GLOSSY, p
    raise TypeError(n)


class Oracle:
    """Scene + Camera of the CPU restatement for one (scene, config, seed)."""

    def __init__(self, scene_data, config, seed=1):
        sd = scene_data
        cam = list(sd.camera_settings.eye) + list(sd.camera_settings.look_at) + list(sd.camera_settings.up) + [
            sd.camera_data.zoom_factor, sd.camera_data.view_plane_distance, sd.camera_data.focal_distance,
            sd.camera_data.lens_radius]
        kinds, sp, mk, mp = [], [], [], []
        meshes = [s for s in sd.shapes if type(s).__name__ == "MeshData"]
        for s in sd.shapes:
            p = [0.0] * 8
            if type(s).__name__ == "MeshData":
                continue  # extension: appended after all analytic shapes, see below
            if type(s).__name__ == "SphereData":
                kinds.append(SHAPE_SPHERE)
                p[0:3] = s.center
                p[3] = s.radius
                p[4] = 1.0 if s.invert else 0.0
            else:
                kinds.append(SHAPE_PLANE)
                p[0:3] = s.point
                p[3:6] = s.normal
            sp += p
            k, q = material_params(s.material)
            mk.append(k)
            mp += q
        n = len(kinds)
        self.width = sd.output_settings.image_width
        self.height = sd.output_settings.image_height
        self.n = config.sample_root
        self.N = self.n * self.n
        self.D = config.max_trace_depth
        self.seed = seed
        cam_a, cam_p = _d(cam)
        bg_a, bg_p = _d(sd.background)
        sp_a, sp_p = _d(sp if sp else [0.0])
        mp_a, mp_p = _d(mp if mp else [0.0])
        kinds_a = np.ascontiguousarray(kinds if kinds else [0], dtype=np.int32)
        mk_a = np.ascontiguousarray(mk if mk else [0], dtype=np.int32)
        self._h = lib.fxo_ctx_create(cam_p, self.width, self.height, sd.output_settings.pixel_size, bg_p, n,
                                     kinds_a.ctypes.data_as(_ip), sp_p, mk_a.ctypes.data_as(_ip), mp_p, self.n,
                                     self.D, C.c_uint64(seed))
        if not self._h:
            raise ValueError("fxo_ctx_create failed (bad arguments)")
        for m in meshes:  # extension (no reference counterpart): brute-force triangles
            v = np.ascontiguousarray(m.vertices, dtype=np.float64).reshape(-1, 3)
            t = np.ascontiguousarray(m.triangles, dtype=np.uint32).reshape(-1, 3)
            k, q = material_params(m.material)
            q_a, q_p = _d(q)
            rc = lib.fxo_ctx_add_mesh(self._h, v.ctypes.data_as(_dp), len(v), t.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      len(t), k, q_p)
            if rc != 0:
                raise ValueError("fxo_ctx_add_mesh failed")

    def close(self):
        if getattr(self, "_h", None):
            lib.fxo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def render_rows(self, row_start, row_end, threads=1):
        n = row_end - row_start + 1
        out = np.empty((n, self.width, 3), dtype=np.float64)
        rc = lib.fxo_render_rows(self._h, row_start, row_end, out.ctypes.data_as(_dp), threads)
        if rc != 0:
            raise ValueError(f"fxo_render_rows({row_start},{row_end}) failed")
        return out

    def render_row_list(self, rows, threads=1):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        out = np.empty((len(rows), self.width, 3), dtype=np.float64)
        rc = lib.fxo_render_row_list(self._h, rows.ctypes.data_as(_ip), len(rows), out.ctypes.data_as(_dp), threads)
        if rc != 0:
            raise ValueError("fxo_render_row_list failed")
        return out

    def render_frame(self, threads=1):
        return self.render_rows(0, self.height - 1, threads)

    def stats(self, reset=False):
        buf = (C.c_uint64 * 8)()
        lib.fxo_stats(self._h, buf)
        if reset:
            lib.fxo_stats_reset(self._h)
        return dict(zip(STAT_NAMES, [int(x) for x in buf]))

    def pixel_sets(self):
        return np.ctypeslib.as_array(lib.fxo_pixel_sets(self._h), shape=(self.width, self.N, 2)).copy()

    def disc_sets(self):
        return np.ctypeslib.as_array(lib.fxo_disc_sets(self._h), shape=(self.width, self.N, 2)).copy()

    def hemi_sets(self):
        return np.ctypeslib.as_array(lib.fxo_hemi_sets(self._h), shape=(self.width, self.D, self.N, 3)).copy()

    def row_perm(self, row):
        out = np.empty(self.width, dtype=np.int32)
        lib.fxo_row_perm(self._h, row, out.ctypes.data_as(_ip))
        return out

    def camera_basis(self):
        out = np.empty(9)
        lib.fxo_camera_basis(self._h, out.ctypes.data_as(_dp))
        return out.reshape(3, 3)

    def primary_ray(self, row, col, set_index, sample_index):
        o, d = np.empty(3), np.empty(3)
        lib.fxo_primary_ray(self._h, row, col, set_index, sample_index, o.ctypes.data_as(_dp), d.ctypes.data_as(_dp))
        return o, d

    def scene_hit(self, o, d):
        t = C.c_double()
        n, p = np.empty(3), np.empty(3)
        idx = lib.fxo_scene_hit(self._h, vec(*o), vec(*d), C.byref(t), n.ctypes.data_as(_dp), p.ctypes.data_as(_dp))
        return idx, t.value, n, p

    def shade(self, o, d, depth, set_index, sample_index):
        out = np.empty(3)
        lib.fxo_shade(self._h, vec(*o), vec(*d), depth, set_index, sample_index, out.ctypes.data_as(_dp))
        return out


# ---- unit-level helpers ------------------------------------------------------------

def grid(kind_fn, seed, kind, a, b, root):
    out = np.empty((root * root, 2))
    kind_fn(seed, kind, a, b, root, out.ctypes.data_as(_dp))
    return out


def grid_multi_jittered(seed, kind, a, b, root):
    return grid(lib.fxo_grid_multi_jittered, seed, kind, a, b, root)


def grid_correlated_multi_jittered(seed, kind, a, b, root):
    return grid(lib.fxo_grid_correlated_multi_jittered, seed, kind, a, b, root)


def grid_regular(root):
    out = np.empty((root * root, 2))
    lib.fxo_grid_regular(root, out.ctypes.data_as(_dp))
    return out


def grid_jittered(key, root):
    out = np.empty((root * root, 2))
    lib.fxo_grid_jittered(key, root, out.ctypes.data_as(_dp))
    return out


def to_unit_hemi(x, y, e):
    out = np.empty(3)
    lib.fxo_to_unit_hemi(x, y, e, out.ctypes.data_as(_dp))
    return out


def to_poisson_disc(x, y):
    out = np.empty(2)
    lib.fxo_to_poisson_disc(x, y, out.ctypes.data_as(_dp))
    return out


def max_to_one(rgb):
    a = np.array(rgb, dtype=np.float64)
    lib.fxo_max_to_one(a.ctypes.data_as(_dp))
    return a


def bbox_hit(c0, c1, o, d):
    return bool(lib.fxo_bbox_hit(vec(*c0), vec(*c1), vec(*o), vec(*d)))


def sphere_hit(center, radius, invert, o, d):
    t = C.c_double()
    n, p = np.empty(3), np.empty(3)
    ok = lib.fxo_sphere_hit(vec(*center), radius, int(invert), vec(*o), vec(*d), C.byref(t),
                            n.ctypes.data_as(_dp), p.ctypes.data_as(_dp))
    return (t.value, n, p) if ok else None


def plane_hit(point, normal, o, d):
    t = C.c_double()
    n, p = np.empty(3), np.empty(3)
    ok = lib.fxo_plane_hit(vec(*point), vec(*normal), vec(*o), vec(*d), C.byref(t), n.ctypes.data_as(_dp),
                           p.ctypes.data_as(_dp))
    return (t.value, n, p) if ok else None


def sample_f(mat_kind, mat_params, n, wo, hemi=(0, 0, 1), sq=(0, 0)):
    wi, f = np.empty(3), np.empty(3)
    pdf = C.c_double()
    lib.fxo_sample_f(mat_kind, vec(*mat_params), vec(*n), vec(*wo), vec(*hemi), vec(*sq), wi.ctypes.data_as(_dp),
                     C.byref(pdf), f.ctypes.data_as(_dp))
    return wi, pdf.value, f


def shuffle(key, n):
    v = np.arange(n, dtype=np.int32)
    lib.fxo_shuffle(key, v.ctypes.data_as(_ip), n)
    return v


def work_units(height, rows):
    cap = height + 1
    s = (C.c_size_t * cap)()
    e = (C.c_size_t * cap)()
    n = lib.fxo_work_units(height, rows, s, e, cap)
    if n == C.c_size_t(-1).value:
        raise ValueError("rows_per_work_unit == 0")
    return [(int(s[i]), int(e[i])) for i in range(n)]


def ppm_quantize(c):
    return int(lib.fxo_ppm_quantize(c))


def write_ppm(path, rgb, rows_present=None):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w, _ = rgb.shape
    rp = None
    if rows_present is not None:
        rp_a = np.ascontiguousarray(rows_present, dtype=np.uint8)
        rp = rp_a.ctypes.data_as(C.POINTER(C.c_uint8))
    if lib.fxo_write_ppm(path.encode(), rgb.ctypes.data_as(_dp), w, h, rp) != 0:
        raise IOError(path)
