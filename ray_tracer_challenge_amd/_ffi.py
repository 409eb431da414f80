"""ctypes binding of `include/rtc_api.h` (librtc_hip.so).  No CPU fallback: if the library is
missing, or there is no GPU, every entry point here raises."""
import ctypes as C
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTC_LIB") or os.path.join(_HERE, "librtc_hip.so")      # RTC_LIB: development builds (tools/build_variant.sh)
_lib = None
_P = C.c_void_p

# (field name, kind) in the exact order of rtc_scene_desc; kind 'n' = int32 count,
# 'n64' = int64 count, 'p' = pointer to the array of the same name in CompiledScene.arrays
_DESC_FIELDS = [
    ("n_leaves", "n"), ("leaf_kind", "p"), ("leaf_xform", "p"), ("leaf_material", "p"), ("leaf_tri", "p"),
    ("leaf_minmax", "p"), ("leaf_closed", "p"),
    ("n_xforms", "n"), ("xform", "p"),
    ("n_tris", "n"), ("tri_geom", "p"), ("tri_norm", "p"), ("tri_leaf", "p"),
    ("n_items", "n"), ("item_prim", "p"), ("item_lo", "p"), ("item_hi", "p"),
    ("n_unbounded", "n"), ("unbounded_prim", "p"),
    ("n_csg_roots", "n"), ("csg_root", "p"), ("n_csg_nodes", "n"), ("csg_node", "p"), ("csg_bbox", "p"),
    ("n_materials", "n"), ("mat", "p"), ("mat_shadow", "p"), ("mat_program", "p"),
    ("n_pat_nodes", "n"), ("pat_type", "p"), ("pat_inv", "p"), ("pat_ab", "p"), ("pat_child", "p"),
    ("pat_param", "p"), ("pat_colors", "p"), ("pat_tex", "p"),
    ("n_programs", "n"), ("prog_off", "p"), ("prog_len", "p"), ("n_instr", "n"), ("instr", "p"),
    ("n_textures", "n"), ("tex_off", "p"), ("tex_w", "p"), ("tex_h", "p"), ("tex_scale", "p"),
    ("tex_bytes", "n64"), ("tex_data", "p"),
    ("n_lights", "n"), ("light_type", "p"), ("light_geom", "p"), ("light_steps", "p"), ("light_intensity", "p"),
    ("n_mesh_tris", "n"), ("mesh_geom", "p"), ("mesh_norm", "p"), ("n_meshes", "n"), ("mesh_range", "p"), ("n_instances", "n"), ("inst", "p"),
    ("build_flags", "flag"),
]
BUILD_ON_DEVICE = 1
_DTYPES = {"tex_off": np.int64, "tex_data": np.uint8}
_F64 = {"leaf_minmax", "xform", "tri_geom", "tri_norm", "item_lo", "item_hi", "csg_bbox", "mat", "pat_inv", "pat_ab",
        "pat_param", "pat_colors", "tex_scale", "light_geom", "light_intensity", "mesh_geom", "mesh_norm"}


class SceneDesc(C.Structure):
    _fields_ = [(n, C.c_int32 if k in ("n", "flag") else (C.c_int64 if k == "n64" else _P)) for n, k in _DESC_FIELDS]


class CameraDesc(C.Structure):
    _fields_ = [("hsize", C.c_int32), ("vsize", C.c_int32), ("half_width", C.c_double),
                ("half_height", C.c_double), ("pixel_size", C.c_double), ("inv_transform", C.c_double * 12)]


class RenderOpts(C.Structure):
    _fields_ = [("precision", C.c_int32), ("max_depth", C.c_int32), ("tile_size", C.c_int32),
                ("flags", C.c_int32), ("shade_eps", C.c_double), ("shade_eps_rel", C.c_double), ("seed", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_int64 * 4), ("work", C.c_int64 * 8), ("kernel_ms", C.c_double), ("shadow_rays_culled", C.c_int64),
                ("wide_node_visits", C.c_int64)]


class SceneInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("bvh_nodes", C.c_int32), ("bvh_depth", C.c_int32),
                ("bvh_max_leaf", C.c_int32), ("device_bytes_fp64", C.c_int64), ("device_bytes_fp32", C.c_int64)]


class LaunchInfo(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("precision", "csg", "far_camera", "replicas", "counting_work", "waves_per_simd",
                                          "workgroup_threads", "workgroups", "n_units", "n_entries", "feedback_stage")]


EXPORTS = ["rtc_scene_last_launch", "rtc_version", "rtc_device_count", "rtc_last_error", "rtc_scene_create", "rtc_scene_destroy",
           "rtc_scene_get_info", "rtc_render", "rtc_render_tiles", "rtc_render_device", "rtc_render_device_noalloc", "rtc_quantize_device",
           "rtc_scatter_tiles_device", "rtc_scatter_id_tiles_device", "rtc_frame_max_device", "rtc_ppm_body_device", "rtc_ppm_size", "rtc_render_ppm",
           "rtc_scene_build_bvh_device", "rtc_scene_update_triangles", "rtc_scene_refit_triangles", "rtc_scene_update_triangles_device",
           "rtc_host_alloc", "rtc_host_free"]


class RtcError(RuntimeError):
    pass


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (SONAME
    libamdhip64.so.7, found through torch/lib's RPATH under the unversioned name), librtc_hip.so
    asks for libamdhip64.so.7: if ours came first the loader would map BOTH copies and the second
    runtime sees no GPU.  Loading torch's copy first (when torch is installed) makes either
    import order end with a single runtime; without torch the system one in /opt/rocm is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtcError("librtc_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C ray_tracer_challenge_amd/csrc`); there is no CPU fallback")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.rtc_last_error.restype = C.c_char_p
        L.rtc_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(_P)]
        L.rtc_scene_destroy.argtypes = [_P]
        L.rtc_scene_destroy.restype = None
        L.rtc_scene_get_info.argtypes = [_P, C.POINTER(SceneInfo)]
        L.rtc_scene_last_launch.argtypes = [_P, C.POINTER(LaunchInfo)]
        L.rtc_render.argtypes = [_P, C.POINTER(CameraDesc), C.POINTER(RenderOpts), _P, _P, C.POINTER(Stats)]
        L.rtc_render_tiles.argtypes = [_P, C.POINTER(CameraDesc), C.POINTER(RenderOpts), C.c_int32, _P, _P, _P,
                                       C.POINTER(Stats)]
        L.rtc_render_device.argtypes = [_P, C.POINTER(CameraDesc), C.POINTER(RenderOpts), C.c_int32, _P, _P, _P, _P, _P]
        L.rtc_render_device_noalloc.argtypes = L.rtc_render_device.argtypes
        L.rtc_quantize_device.argtypes = [_P, C.c_int32, C.c_int32, C.c_double, _P, _P]
        L.rtc_scatter_tiles_device.argtypes = [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]
        L.rtc_scatter_id_tiles_device.argtypes = L.rtc_scatter_tiles_device.argtypes
        L.rtc_scene_build_bvh_device.argtypes = [_P]
        L.rtc_scene_update_triangles.argtypes = [_P, C.c_int64, _P, _P]
        L.rtc_scene_refit_triangles.argtypes = [_P, C.c_int64, _P, _P]
        L.rtc_scene_update_triangles_device.argtypes = [_P, C.c_int64, _P, _P, C.c_int]
        L.rtc_frame_max_device.argtypes = [_P, C.c_int64, _P, _P]
        L.rtc_host_alloc.argtypes = [C.c_size_t, C.POINTER(_P)]
        L.rtc_host_free.argtypes = [_P]
        L.rtc_host_free.restype = None
        L.rtc_ppm_body_device.argtypes = [_P, C.c_int32, C.c_int32, _P, _P, _P]
        L.rtc_ppm_size.argtypes = [C.c_int32, C.c_int32]
        L.rtc_ppm_size.restype = C.c_int64
        L.rtc_render_ppm.argtypes = [_P, C.POINTER(CameraDesc), C.POINTER(RenderOpts), _P, _P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(Stats)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RtcError("rtc error %d: %s" % (rc, lib().rtc_last_error().decode("utf-8", "replace")))


def device_count():
    return int(lib().rtc_device_count())


class _PinnedPool(object):
    """Result arrays in page-locked host memory (rtc_host_alloc): the 50 MB fp64 frame of a 1080p render crosses PCIe in
    0.9 ms into pinned memory, 2.7 ms into touched pageable memory and 5-12 ms into a fresh numpy array (page faults).
    A block goes back to the pool when the last numpy view of it is collected; beyond LIMIT bytes handed out or cached,
    plain numpy memory is used (a caller keeping hundreds of frames must not pin them all)."""
    LIMIT = 1 << 30
    KEEP = 256 << 20

    def __init__(self):
        self.lock = threading.Lock()
        self.free = {}
        self.cached = 0
        self.total = 0

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes < (1 << 16):
            return np.empty(shape, dtype)
        nbytes = (nbytes + 4095) & ~4095
        ptr = None
        with self.lock:
            blocks = self.free.get(nbytes)
            if blocks:
                ptr = blocks.pop()
                self.cached -= nbytes
            elif self.total + nbytes > self.LIMIT:
                return np.empty(shape, dtype)
            else:
                self.total += nbytes
        if ptr is None:
            p = _P()
            if lib().rtc_host_alloc(nbytes, C.byref(p)) != 0 or not p.value:
                with self.lock:
                    self.total -= nbytes
                return np.empty(shape, dtype)
            ptr = p.value
        buf = (C.c_char * nbytes).from_address(ptr)
        weakref.finalize(buf, self._give_back, ptr, nbytes).atexit = False
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def _give_back(self, ptr, nbytes):
        with self.lock:
            if self.cached + nbytes <= self.KEEP:
                self.free.setdefault(nbytes, []).append(ptr)
                self.cached += nbytes
                return
            self.total -= nbytes
        lib().rtc_host_free(_P(ptr))


_pinned = _PinnedPool()


def make_camera(cam):
    c = CameraDesc()
    c.hsize, c.vsize = int(cam.hsize), int(cam.vsize)
    c.half_width, c.half_height, c.pixel_size = float(cam.half_width), float(cam.half_height), float(cam.pixel_size)
    if getattr(cam, "inv_transform", None) is not None:       # a camera that states its inverse itself (tests: exact rays)
        inv = np.asarray(cam.inv_transform, dtype=np.float64).reshape(-1)[:12]
    else:
        t = np.asarray(getattr(cam.transform, "arr", cam.transform), dtype=np.float64)
        inv = np.linalg.inv(t)[:3, :].reshape(12)      # inverse(cam.transform), renderer.py:747-748
    for i in range(12):
        c.inv_transform[i] = inv[i]
    return c


def make_opts(precision="fp64", max_depth=5, tile_size=32, shade_eps=0.0, seed=0, count_work=False, shade_eps_rel=-1.0,
              feedback=True):
    bits = {"fp64": 64, "fp32": 32, 64: 64, 32: 32}.get(precision)
    if bits is None:
        raise ValueError("precision must be 'fp64' or 'fp32'")
    return RenderOpts(bits, int(max_depth), int(tile_size), (1 if count_work else 0) | (0 if feedback else 8), float(shade_eps), float(shade_eps_rel), int(seed))


def make_desc(cs, build_flags=0):
    """CompiledScene -> (SceneDesc, keepalive list of arrays)."""
    d = SceneDesc()
    keep = []
    for name, kind in _DESC_FIELDS:
        if kind == "flag":
            setattr(d, name, int(build_flags))
        elif kind in ("n", "n64"):
            setattr(d, name, int(cs.counts[name]))
        else:
            want = _DTYPES.get(name, np.float64 if name in _F64 else np.int32)
            a = np.ascontiguousarray(cs.arrays[name], dtype=want)
            keep.append(a)
            setattr(d, name, a.ctypes.data_as(_P))
    return d, keep


class Scene(object):
    """A compiled scene resident on one GPU (`rtc_scene*`)."""

    def __init__(self, cs, device=0, build_on_device=False):
        """build_on_device: the triangle BVH is built on the GPU (~2 ms instead of tens; frames ~1.1 x slower): for a scene
        that is rendered once (rtc_scene_desc::build_flags)."""
        self.handle = _P()
        self.cs = cs
        d, keep = make_desc(cs, BUILD_ON_DEVICE if build_on_device else 0)
        _check(lib().rtc_scene_create(C.byref(d), int(device), C.byref(self.handle)))
        self.device = device

    def close(self):
        if self.handle:
            lib().rtc_scene_destroy(self.handle)
            self.handle = _P()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def build_bvh_on_device(self):
        """Replace the host-built triangle BVH by one built on the GPU (rtc_scene_build_bvh_device)."""
        _check(lib().rtc_scene_build_bvh_device(self.handle))

    def update_triangles(self, tri_geom, tri_norm, refit=False):
        """New world-space triangle rows (the compiled scene's tri_geom / tri_norm arrays), then a GPU rebuild of the
        triangle BVH, or with refit=True only new boxes on the tree the scene has (rtc_scene_refit_triangles)."""
        g = np.ascontiguousarray(tri_geom, np.float64)
        nn = np.ascontiguousarray(tri_norm, np.float64)
        fn = lib().rtc_scene_refit_triangles if refit else lib().rtc_scene_update_triangles
        _check(fn(self.handle, g.size // 12, g.ctypes.data_as(_P), nn.ctypes.data_as(_P)))

    def update_triangles_device(self, geom_ptr, norm_ptr, n_tris, refit=False):
        """The same with the rows already on this scene's GPU: device pointers (e.g. torch tensors' data_ptr()) to
        n_tris x 12 fp64 each; the caller synchronises whatever wrote them."""
        _check(lib().rtc_scene_update_triangles_device(self.handle, int(n_tris), C.c_void_p(geom_ptr), C.c_void_p(norm_ptr), 1 if refit else 0))

    def info(self):
        out = SceneInfo()
        _check(lib().rtc_scene_get_info(self.handle, C.byref(out)))
        return {k: getattr(out, k) for k, _ in SceneInfo._fields_}

    def last_launch(self):
        """Which trace kernel variant the most recent frame of this handle was launched as, and its grid (reporting only)."""
        out = LaunchInfo()
        _check(lib().rtc_scene_last_launch(self.handle, C.byref(out)))
        d = {k: getattr(out, k) for k, _ in LaunchInfo._fields_}
        d["kernel"] = "trace_kernel<%s, %s, %d, %s, %s, %d>" % ("double" if d["precision"] == 64 else "float", "true" if d["counting_work"] else "false", d["csg"],
                                                               "true" if d["far_camera"] else "false", "true" if d["replicas"] else "false", d["waves_per_simd"])
        return d

    def render(self, cam, precision="fp64", max_depth=5, want_ids=False, stats=None, shade_eps=0.0, seed=0,
               count_work=False, feedback=True):
        """-> (rgb (hsize, vsize, 3) f64, ids (hsize, vsize) i32 or None)."""
        c, o, st = make_camera(cam), make_opts(precision, max_depth, 32, shade_eps, seed, count_work, feedback=feedback), Stats()
        rgb = _pinned.empty((c.hsize, c.vsize, 3), np.float64)       # every element is written by the call
        ids = _pinned.empty((c.hsize, c.vsize), np.int32) if want_ids else None
        _check(lib().rtc_render(self.handle, C.byref(c), C.byref(o), rgb.ctypes.data_as(_P),
                                ids.ctypes.data_as(_P) if want_ids else None, C.byref(st)))
        _fill_stats(stats, st)
        return rgb, ids

    def render_ppm(self, cam, precision="fp64", max_depth=5, clamped=True, scaled=True, stats=None, seed=0):
        """-> (final_clamped.ppm bytes or None, final.ppm bytes or None) like main.py:343-354, encoded on the GPU."""
        c, o, st = make_camera(cam), make_opts(precision, max_depth, 32, 0.0, seed), Stats()
        size = int(lib().rtc_ppm_size(c.hsize, c.vsize))
        a = _pinned.empty((size,), np.uint8) if clamped else None
        b = _pinned.empty((size,), np.uint8) if scaled else None
        n = C.c_int64(0)
        _check(lib().rtc_render_ppm(self.handle, C.byref(c), C.byref(o), a.ctypes.data_as(_P) if clamped else None,
                                    b.ctypes.data_as(_P) if scaled else None, size, C.byref(n), C.byref(st)))
        _fill_stats(stats, st)
        return (a[:n.value].tobytes() if clamped else None, b[:n.value].tobytes() if scaled else None)

    def render_tiles(self, cam, tile_ids, precision="fp64", max_depth=5, tile_size=32, want_ids=False, stats=None,
                     shade_eps=0.0, seed=0, feedback=True, count_work=False):
        """Compact tiles -> (rgb (n, T, T, 3) [slot][ty][tx], ids (n, T, T) or None)."""
        c, o, st = make_camera(cam), make_opts(precision, max_depth, tile_size, shade_eps, seed, count_work, feedback=feedback), Stats()
        tiles = np.ascontiguousarray(tile_ids, dtype=np.int32)
        n, T = tiles.size, tile_size
        rgb = _pinned.empty((n, T, T, 3), np.float64)
        ids = _pinned.empty((n, T, T), np.int32) if want_ids else None
        _check(lib().rtc_render_tiles(self.handle, C.byref(c), C.byref(o), n, tiles.ctypes.data_as(_P),
                                      rgb.ctypes.data_as(_P), ids.ctypes.data_as(_P) if want_ids else None, C.byref(st)))
        _fill_stats(stats, st)
        return rgb, ids

    def render_device(self, cam, rgb_ptr, ids_ptr=None, rays_ptr=None, tile_ids_ptr=None, n_tiles=0, precision="fp64",
                      max_depth=5, tile_size=32, shade_eps=0.0, seed=0, stream=None, count_work=False, feedback=True, noalloc=False):
        """Asynchronous launch on device pointers (ints, e.g. torch `tensor.data_ptr()`); noalloc=True goes through
        rtc_render_device_noalloc (launch-order buffers in the library's own device array, no allocation)."""
        c, o = make_camera(cam), make_opts(precision, max_depth, tile_size, shade_eps, seed, count_work, feedback=feedback)
        fn = lib().rtc_render_device_noalloc if noalloc else lib().rtc_render_device
        _check(fn(self.handle, C.byref(c), C.byref(o), int(n_tiles), _P(tile_ids_ptr or None),
                                       _P(rgb_ptr), _P(ids_ptr or None), _P(rays_ptr or None), _P(stream or None)))


def scatter_tiles_device(tiles_ptr, tile_ids_ptr, n_slots, tile_size, hsize, vsize, frame_ptr, stream=None):
    """Compact tiles in HBM -> the [x][y][c] canvas in HBM (device pointers as ints)."""
    _check(lib().rtc_scatter_tiles_device(_P(tiles_ptr), _P(tile_ids_ptr), int(n_slots), int(tile_size), int(hsize), int(vsize),
                                          _P(frame_ptr), _P(stream or None)))


def scatter_id_tiles_device(tiles_ptr, tile_ids_ptr, n_slots, tile_size, hsize, vsize, frame_ptr, stream=None):
    """Compact hit-id tiles in HBM -> the [x][y] id map in HBM (device pointers as ints)."""
    _check(lib().rtc_scatter_id_tiles_device(_P(tiles_ptr), _P(tile_ids_ptr), int(n_slots), int(tile_size), int(hsize), int(vsize),
                                             _P(frame_ptr), _P(stream or None)))


def _fill_stats(stats, st):
    if stats is not None:
        stats["rays"] = [int(v) for v in st.rays]
        stats["work"] = [int(v) for v in st.work]
        stats["kernel_ms"] = float(st.kernel_ms)
        stats["shadow_rays_culled"] = int(st.shadow_rays_culled)
        stats["wide_node_visits"] = int(st.wide_node_visits)


def tiles_for(hsize, vsize, tile_size=32):
    tx, ty = (hsize + tile_size - 1) // tile_size, (vsize + tile_size - 1) // tile_size
    return tx, ty


def scatter_tiles(dst_rgb, dst_ids, tile_ids, rgb_tiles, id_tiles, tile_size=32):
    """Un-permute compact tiles ([slot][ty][tx]) into the [x][y] frame."""
    hsize, vsize = dst_rgb.shape[0], dst_rgb.shape[1]
    tiles_x, tiles_y = tiles_for(hsize, vsize, tile_size)
    T = tile_size
    tiles = np.asarray(tile_ids).astype(np.int64).reshape(-1)
    ix, iy = tiles % tiles_x, tiles // tiles_x

    def place(dst, src, tail):
        # the frame as [tile x][tx][tile y][ty]...; frames whose sides are not multiples of T go through a padded copy
        whole = hsize == tiles_x * T and vsize == tiles_y * T and dst.flags.c_contiguous
        full = dst if whole else np.zeros((tiles_x * T, tiles_y * T) + tail, dst.dtype)
        if not whole:
            full[:hsize, :vsize] = dst
        full.reshape((tiles_x, T, tiles_y, T) + tail)[ix, :, iy, :] = np.swapaxes(src, 1, 2)
        if not whole:
            dst[...] = full[:hsize, :vsize]

    place(dst_rgb, rgb_tiles, (3,))
    if dst_ids is not None and id_tiles is not None:
        place(dst_ids, id_tiles, ())


class DeviceScenes(object):
    """One compiled scene resident on several GPUs -- one `rtc_scene` per device, created and uploaded ONCE -- for any
    number of frames: tile i of a frame goes to devices[i % n] (one host thread per GPU), host-side gather.  The
    single-process counterpart of the one-process-per-GPU form in dist.py, for callers without a launcher."""

    def __init__(self, cs, devices, build_on_device=False):
        self.devices = list(devices)
        self.scenes = []
        try:
            for d in self.devices:
                self.scenes.append(Scene(cs, device=d, build_on_device=build_on_device))
        except Exception:
            self.close()
            raise

    def close(self):
        for sc in self.scenes:
            sc.close()
        self.scenes = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def render(self, cam, precision="fp64", max_depth=5, want_ids=False, stats=None, tile_size=32):
        if len(self.scenes) == 1:
            return self.scenes[0].render(cam, precision, max_depth, want_ids, stats)
        tx, ty = tiles_for(cam.hsize, cam.vsize, tile_size)
        all_tiles = np.arange(tx * ty, dtype=np.int32)
        rgb = np.zeros((cam.hsize, cam.vsize, 3), np.float64)
        ids = np.full((cam.hsize, cam.vsize), -1, np.int32) if want_ids else None
        shares = [(sc, all_tiles[k::len(self.scenes)]) for k, sc in enumerate(self.scenes)]
        shares = [(sc, mine) for sc, mine in shares if mine.size]

        def one(share):
            sc, mine = share
            st = {}
            r, i = sc.render_tiles(cam, mine, precision, max_depth, tile_size, want_ids, st)
            return mine, r, i, st

        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(shares)) as pool:      # the C calls release the GIL
            results = list(pool.map(one, shares))
        total, culled = [0, 0, 0, 0], 0
        for mine, r, i, st in results:
            scatter_tiles(rgb, ids, mine, r, i, tile_size)
            total = [a + b for a, b in zip(total, st["rays"])]
            culled += st["shadow_rays_culled"]
        if stats is not None:
            stats["rays"] = total
            stats["shadow_rays_culled"] = culled
        return rgb, ids


def render_on_devices(cs, cam, devices, precision="fp64", max_depth=5, want_ids=False, stats=None, tile_size=32, scenes=None):
    """One frame on several GPUs of one process.  `scenes` (a DeviceScenes the caller keeps) renders it on scenes that
    are already resident; without it the scene is created on every device for this call and released after it."""
    if scenes is not None:
        return scenes.render(cam, precision, max_depth, want_ids, stats, tile_size)
    with DeviceScenes(cs, devices, build_on_device=True) as ds:      # one frame on this scene: the tree that is quickest to build
        return ds.render(cam, precision, max_depth, want_ids, stats, tile_size)
