"""TEST INFRASTRUCTURE ONLY.  ctypes front of `librtc_oracle.so` (see rtc_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from .export import OracleScene, _a

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librtc_oracle.so")
_lib = None

_P = C.c_void_p


class _Scene(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("kind", _P), ("parent", _P), ("inv", _P), ("material", _P),
        ("fparam", _P), ("child_off", _P), ("child_cnt", _P), ("child_idx", _P), ("csg_op", _P),
        ("bbox", _P), ("tri_idx", _P), ("tri", _P), ("leaf_id", _P),
        ("n_top", C.c_int32), ("top", _P),
        ("n_materials", C.c_int32), ("mat", _P), ("mat_shadow", _P), ("mat_pattern", _P),
        ("n_patterns", C.c_int32), ("pat_type", _P), ("pat_inv", _P), ("pat_ab", _P),
        ("pat_child", _P), ("pat_param", _P), ("pat_colors", _P), ("pat_tex", _P),
        ("n_textures", C.c_int32), ("tex_off", _P), ("tex_w", _P), ("tex_h", _P),
        ("tex_scale", _P), ("tex_data", _P),
        ("n_lights", C.c_int32), ("light_type", _P), ("light_geom", _P), ("light_steps", _P),
        ("light_intensity", _P),
    ]


class _Camera(C.Structure):
    _fields_ = [("hsize", C.c_int32), ("vsize", C.c_int32), ("half_width", C.c_double),
                ("half_height", C.c_double), ("pixel_size", C.c_double),
                ("inv_transform", C.c_double * 16)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rtc_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librtc_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_render_pixels.restype = C.c_int
        _lib.orc_pnoise3.restype = C.c_double
        _lib.orc_pnoise3.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
    return _lib


def pnoise3(x, y, z, octaves=1):
    """noise.pnoise3 as the reference calls it (shapes.py:1916), restated in rtc_oracle.c -- parity unpinned."""
    return float(lib().orc_pnoise3(float(x), float(y), float(z), int(octaves)))


def _ptr(a):
    return a.ctypes.data_as(_P)


def _v3(x):
    return np.ascontiguousarray(_a(x, 3), dtype=np.float64)


def make_camera(cam):
    c = _Camera()
    c.hsize, c.vsize = int(cam.hsize), int(cam.vsize)
    c.half_width, c.half_height, c.pixel_size = float(cam.half_width), float(cam.half_height), float(cam.pixel_size)
    if getattr(cam, "inv_transform", None) is not None:
        inv = np.asarray(cam.inv_transform, dtype=np.float64).reshape(16)
    else:
        inv = np.linalg.inv(_a(cam.transform)).reshape(16)
    for i in range(16):
        c.inv_transform[i] = inv[i]
    return c


class Oracle(object):
    """The reference's trace loop on the CPU for `world` (our objects or the reference's)."""

    def __init__(self, world):
        self.ex = ex = OracleScene(world)
        a = ex.a
        s = _Scene()
        s.n_nodes = len(ex.nodes)
        for k in ("kind", "parent", "inv", "material", "fparam", "child_off", "child_cnt", "child_idx",
                  "csg_op", "bbox", "tri_idx", "tri", "leaf_id", "top", "mat", "mat_shadow", "mat_pattern",
                  "pat_type", "pat_inv", "pat_ab", "pat_child", "pat_param", "pat_colors", "pat_tex",
                  "tex_off", "tex_w", "tex_h", "tex_scale", "tex_data", "light_type", "light_geom",
                  "light_steps", "light_intensity"):
            a[k] = np.ascontiguousarray(a[k])
            setattr(s, k, _ptr(a[k]))
        s.n_top = len(ex.top)
        s.n_materials = len(ex.materials)
        s.n_patterns = len(ex.patterns)
        s.n_textures = len(ex.textures)
        s.n_lights = ex.n_lights
        self.s = s
        self.n_leaves = len(ex.leaf_nodes)

    # ---- frame-level -------------------------------------------------------------------
    def render_pixels(self, cam, px, py, remaining=5, threads=None):
        """-> rgb (n,3) f64, hit leaf id (n,) i32, hit t (n,) f64, rays (4,) i64."""
        px = np.ascontiguousarray(px, dtype=np.int32)
        py = np.ascontiguousarray(py, dtype=np.int32)
        n = px.size
        rgb = np.zeros((n, 3), np.float64)
        leaf = np.zeros(n, np.int32)
        t = np.zeros(n, np.float64)
        rays = np.zeros(4, np.int64)
        c = make_camera(cam)
        if threads is None:
            threads = min(16, os.cpu_count() or 1)
        rc = lib().orc_render_pixels(C.byref(self.s), C.byref(c), C.c_int64(n), _ptr(px), _ptr(py),
                                     int(remaining), int(threads), _ptr(rgb), _ptr(leaf), _ptr(t), _ptr(rays))
        if rc != 0:
            raise RuntimeError("oracle: jittered area lights are not reproducible in the reference; "
                               "set jitter false for parity runs")
        return rgb, leaf, t, rays

    def render(self, cam, remaining=5, threads=None):
        """Whole frame -> (hsize, vsize, 3) f64, (hsize, vsize) i32 ids, rays."""
        xs, ys = np.meshgrid(np.arange(cam.hsize), np.arange(cam.vsize), indexing="ij")
        rgb, leaf, t, rays = self.render_pixels(cam, xs.ravel(), ys.ravel(), remaining, threads)
        return rgb.reshape(cam.hsize, cam.vsize, 3), leaf.reshape(cam.hsize, cam.vsize), rays

    # ---- known-answer-test level ---------------------------------------------------------
    def ray_for_pixel(self, cam, px, py):
        c = make_camera(cam)
        o, d = np.zeros(3), np.zeros(3)
        lib().orc_ray_for_pixel(C.byref(c), int(px), int(py), _ptr(o), _ptr(d))
        return o, d

    def _xs(self, fn, extra, o, d, cap=256):
        t = np.zeros(cap)
        node = np.zeros(cap, np.int32)
        o, d = _v3(o), _v3(d)
        n = fn(C.byref(self.s), *extra, _ptr(o), _ptr(d), cap, _ptr(t), _ptr(node))
        return t[:n].copy(), node[:n].copy()

    def intersect_world(self, o, d, cap=256):
        return self._xs(lib().orc_intersect_world, (), o, d, cap)

    def intersect(self, obj, o, d):
        """shapes.intersect(obj, ray): `obj` is a scene object or a node index."""
        node = obj if isinstance(obj, int) else self.ex.node_of(obj)
        return self._xs(lib().orc_intersect_node, (int(node),), o, d)

    def color_at(self, o, d, remaining=5):
        out = np.zeros(3)
        o, d = _v3(o), _v3(d)
        lib().orc_color_at(C.byref(self.s), _ptr(o), _ptr(d), int(remaining), _ptr(out))
        return out

    def normal_at(self, obj, p, u=0.0, v=0.0):
        out = np.zeros(3)
        p = _v3(p)
        lib().orc_normal_at(C.byref(self.s), int(self.ex.node_of(obj)), _ptr(p), C.c_double(u), C.c_double(v), _ptr(out))
        return out

    def pattern_at_shape(self, pattern, obj, p):
        out = np.zeros(3)
        p = _v3(p)
        lib().orc_pattern_at_shape(C.byref(self.s), int(self.ex.pattern_of(pattern)), int(self.ex.node_of(obj)),
                                   _ptr(p), _ptr(out))
        return out

    def lighting(self, material, obj, light_index, p, eyev, normalv, intensity=1.0):
        out = np.zeros(3)
        p, e, n = _v3(p), _v3(eyev), _v3(normalv)
        lib().orc_lighting(C.byref(self.s), int(self.ex.material_of(material)), int(self.ex.node_of(obj)),
                           int(light_index), _ptr(p), _ptr(e), _ptr(n), C.c_double(intensity), _ptr(out))
        return out

    def is_shadowed(self, light_position, p):
        lp, p = _v3(light_position), _v3(p)
        return bool(lib().orc_is_shadowed(C.byref(self.s), _ptr(lp), _ptr(p)))

    def prepare_computations(self, o, d, k):
        """comps of the k-th entry of intersect_world(ray) as a dict."""
        out = np.zeros(23)
        o, d = _v3(o), _v3(d)
        node = lib().orc_prepare_computations(C.byref(self.s), _ptr(o), _ptr(d), int(k), _ptr(out))
        if node < 0:
            raise IndexError(k)
        return {"t": out[0], "point": out[1:4], "eyev": out[4:7], "normalv": out[7:10], "inside": bool(out[10]),
                "reflectv": out[11:14], "over_point": out[14:17], "under_point": out[17:20],
                "n1": out[20], "n2": out[21], "schlick": out[22], "node": node}
