"""`render(camera, world) -> canvas`: the drop-in boundary of the reference's `renderer.py`.

What stays on the host, with the reference's names (renderer.py:18-21, 613-713, 895-987):
`World`, `Camera` (+ `_compute_sizes`), `view_transform`, `PointLight`, `AreaLight`.
What the reference does per pixel in Python (`ray_for_pixel` -> `color_at` -> `shade_hit`
-> `lighting` / `is_shadowed` / `reflected_color` / `refracted_color` / `schlick`,
renderer.py:98-611, 715-751, 990-1090) is NOT here: `render` and `render_multi` compile
the scene (`compile.compile_scene`) and launch the HIP trace kernel through the C ABI
(`_ffi`, `include/rtc_api.h`).  There is no CPU fallback: without the HIP library or a GPU
these two functions raise.

Both accept either this package's objects or the reference's own `Camera` / `World` /
shape objects (they are read by attribute name only), so in the reference's `main.py`
the only change needed is the import of `render` / `render_multi`.
"""
import numpy as np

from .canvas import Canvas, color
from .matrix import Matrix, matrix4x4identity, translation
from .vector import _arr, cross, normalize, point, vector

EPSILON = 0.000001
BLOCK_SIZE = 2


class World(object):
    def __init__(self, lights, contains):
        self.lights = lights
        self.contains = contains


def world():
    return World(None, [])


def default_world():
    from . import shapes
    light = point_light(point(-10, 10, -10), color(1, 1, 1))
    s1 = shapes.sphere()
    s1.material.color = color(0.8, 1.0, 0.6)
    s1.material.diffuse = np.float64(0.7)
    s1.material.specular = np.float64(0.2)
    s2 = shapes.sphere()
    from .matrix import scaling
    s2.transform = scaling(0.5, 0.5, 0.5)
    return World([light], [s1, s2])


def view_transform(fr, to, up):
    forward = normalize(to - fr)
    left = cross(forward, normalize(up))
    true_up = cross(left, forward)
    orientation = np.eye(4)
    orientation[0, :3] = left.arr[:3]
    orientation[1, :3] = true_up.arr[:3]
    orientation[2, :3] = -forward.arr[:3]
    return Matrix(orientation) * translation(-fr[0], -fr[1], -fr[2])


def view_transform_from_yaml(fr, to, up):
    return view_transform(point(*fr), point(*to), vector(*up))


class Camera(object):
    def __init__(self, hsize, vsize, field_of_view, xform=None):
        self.hsize = hsize
        self.vsize = vsize
        self.field_of_view = float(field_of_view)
        self.transform = matrix4x4identity() if xform is None else xform
        self.half_width, self.half_height, self.pixel_size = self._compute_sizes()

    @classmethod
    def from_yaml(cls, obj):
        return cls(hsize=obj["width"], vsize=obj["height"], field_of_view=obj["field-of-view"],
                   xform=view_transform_from_yaml(obj["from"], obj["to"], obj["up"]))

    def _compute_sizes(self):
        half_view = np.tan(self.field_of_view / 2)
        aspect = float(self.hsize) / float(self.vsize)
        if aspect >= 1:
            half_width, half_height = half_view, half_view / aspect
        else:
            half_width, half_height = half_view * aspect, half_view
        return half_width, half_height, half_width * 2 / self.hsize

    def resized(self, hsize, vsize):
        """Same view at another resolution (the 1920x1080 re-targets of SURVEY.md 8d)."""
        return Camera(hsize, vsize, self.field_of_view, self.transform)


def camera(hsize, vsize, field_of_view):
    return Camera(hsize, vsize, field_of_view)


class Light(object):
    def __init__(self, intensity):
        self.intensity = intensity

    @classmethod
    def from_yaml(cls, obj):
        if "at" in obj:
            return PointLight.from_yaml(obj)
        if "corner" in obj:
            return AreaLight.from_yaml(obj)


class PointLight(Light):
    def __init__(self, position, intensity):
        Light.__init__(self, intensity)
        self.position = position
        self.samples = 1

    @classmethod
    def from_yaml(cls, obj):
        return cls(position=point(*obj["at"]), intensity=color(*obj["intensity"]))


class AreaLight(Light):
    """Rectangular light sampled on a usteps x vsteps grid (reference renderer.py:925-973).

    With `jitter` false every cell is sampled at its centre and the result is the
    reference's, sample for sample.  With `jitter` true the reference draws from Python's
    global RNG in a process-timing-dependent order (not reproducible even against itself);
    the kernels use a counter-based hash of (pixel, bounce, light, cell) seeded by
    `random_seed` instead, which matches it in distribution only.
    """

    def __init__(self, corner, full_uvec, usteps, full_vvec, vsteps, jitter, intensity, random_seed=0):
        Light.__init__(self, intensity)
        self.corner = corner
        self.uvec = full_uvec / usteps
        self.usteps = usteps
        self.vvec = full_vvec / vsteps
        self.vsteps = vsteps
        self.jitter = jitter
        self.samples = usteps * vsteps
        self.random_seed = random_seed

    @classmethod
    def from_yaml(cls, obj):
        return cls(corner=point(*obj["corner"]), full_uvec=vector(*obj["uvec"]),
                   full_vvec=vector(*obj["vvec"]), usteps=int(obj["usteps"]),
                   vsteps=int(obj["vsteps"]), jitter=obj["jitter"], intensity=color(*obj["intensity"]))


def point_light(position, intensity):
    return PointLight(position, intensity)


# --------------------------------------------------------------------------------------
# the boundary
# --------------------------------------------------------------------------------------
def render(cam, world, precision="fp64", max_depth=5, device=0, return_ids=False, stats=None):
    """Trace every pixel of `cam` on one GPU; returns a `Canvas` (reference renderer.py:753-775).

    Keyword extras (not in the reference): `precision` 'fp64' | 'fp32', `max_depth`
    (the reference's `remaining=5`), `device`, `return_ids` (also return the `(hsize,
    vsize)` int32 map of hit leaf ids, -1 = miss), `stats` (dict filled with ray counts).
    """
    from .compile import compile_scene
    from . import _ffi
    cs = compile_scene(world)
    # one frame on this scene (the reference's use: main.py renders every scene once): the triangle BVH is built on the GPU,
    # ~2 ms instead of the host SAH build's tens (device_scenes() keeps a world resident with the SAH tree for many frames)
    with _ffi.Scene(cs, device=device, build_on_device=True) as scene:
        rgb, ids = scene.render(cam, precision=precision, max_depth=max_depth,
                                want_ids=return_ids, stats=stats)
    image = Canvas(cam.hsize, cam.vsize, rgb)
    return (image, ids) if return_ids else image


def render_ppm(cam, world, precision="fp64", max_depth=5, clamped=True, scaled=True, device=0):
    """main.py:343-354 in one call: render, then `construct_ppm(ca)` (final_clamped.ppm) and
    `construct_ppm(ca, False)` (final.ppm, scaled by the brightest component) -- traced AND encoded on
    the GPU, so only the two P6 files cross PCIe.  Returns (clamped_bytes | None, scaled_bytes | None)."""
    from .compile import compile_scene
    from . import _ffi
    with _ffi.Scene(compile_scene(world), device=device, build_on_device=True) as sc:
        return sc.render_ppm(cam, precision=precision, max_depth=max_depth, clamped=clamped, scaled=scaled)


def render_multi(cam, world, num_threads=4, timeout_in_seconds=30, file_path="./", clamping=False,
                 precision="fp64", max_depth=5, devices=None, return_ids=False, stats=None, scenes=None):
    """Same image as `render`, sharded over GPUs (reference renderer.py:795-835).

    Returns the raw `(hsize, vsize, 3)` fp64 ndarray like the reference.  `num_threads`
    (the reference's pool size) caps the number of GPUs used; `devices` overrides it.
    `timeout_in_seconds`, `file_path` and `clamping` only steer the reference's periodic
    partial-image dumps (renderer.py:820-832); a frame here takes milliseconds, so no
    partial image is ever due and they are accepted and ignored.  Any image size is
    accepted (the reference's 2x2 tiling fails on odd sizes; `render` does not).
    A caller that renders many frames of one world (several cameras, a fly-through) creates
    `scenes = device_scenes(world, n)` once and passes it to every call: the world is then compiled
    and uploaded to each GPU once, not once per frame (`world` is ignored in those calls).
    """
    from .compile import compile_scene
    from . import _ffi
    if scenes is not None:
        rgb, ids = scenes.render(cam, precision, max_depth, return_ids, stats)
        return (rgb, ids) if return_ids else rgb
    n_avail = _ffi.device_count()
    if n_avail <= 0:
        raise RuntimeError("render_multi needs at least one GPU (rtc_device_count() = %d)" % n_avail)
    n = devices if devices is not None else max(1, min(int(num_threads), n_avail))
    if n > n_avail:
        raise RuntimeError("asked for %d GPUs, %d visible" % (n, n_avail))
    cs = compile_scene(world)
    rgb, ids = _ffi.render_on_devices(cs, cam, list(range(n)), precision=precision,
                                      max_depth=max_depth, want_ids=return_ids, stats=stats)
    return (rgb, ids) if return_ids else rgb


def device_scenes(world, devices=None):
    """`world` compiled once and resident on `devices` GPUs (an int = the first n, default all): a context manager to
    pass to `render_multi(..., scenes=...)` for many frames of the same world."""
    from .compile import compile_scene
    from . import _ffi
    n_avail = _ffi.device_count()
    if n_avail <= 0:
        raise RuntimeError("device_scenes needs at least one GPU (rtc_device_count() = %d)" % n_avail)
    devs = list(range(n_avail if devices is None else devices)) if not isinstance(devices, (list, tuple)) else list(devices)
    return _ffi.DeviceScenes(compile_scene(world), devs)
