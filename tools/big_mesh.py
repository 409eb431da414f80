"""Scenes far larger than any fixture: a bumpy height field of 2 N^2 triangles written straight into the compiled arrays
(no Python object per triangle), created with the tree built on the GPU, rendered at 1080p in both precisions, and -- for
sizes the host's SAH build finishes -- compared with the frame on the host-built tree.  Usage: big_mesh.py N [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ray_tracer_challenge_amd import _ffi, shapes as S
from ray_tracer_challenge_amd.canvas import color
from ray_tracer_challenge_amd.compile import K_TRIANGLE, compile_scene
from ray_tracer_challenge_amd.matrix import translation
from ray_tracer_challenge_amd.renderer import Camera, PointLight, World, view_transform
from ray_tracer_challenge_amd.vector import point, vector


def height_field(n):
    """-> compiled scene of 2 n^2 flat triangles over [-4, 4]^2 (one material, mildly reflective) and one light"""
    mat = S.Material(color=color(0.3, 0.6, 0.9), reflective=0.3)
    seed_tri = S.Triangle(mat, translation(0, 0, 0), point(0, 0, 0), point(1, 0, 0), point(0, 0, 1))
    cs = compile_scene(World([PointLight(point(-6, 9, -7), color(1, 1, 1)), PointLight(point(6, 5, -3), color(0.4, 0.4, 0.4))], [seed_tri]))
    u = np.linspace(-4, 4, n + 1)
    x, z = np.meshgrid(u, u, indexing="ij")
    y = 0.35 * np.sin(2.1 * x) * np.cos(1.7 * z) + 0.05 * np.sin(9 * x + 4 * z)
    P = np.stack([x, y, z], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    p1 = np.concatenate([a.reshape(-1, 3), a.reshape(-1, 3)])
    p2 = np.concatenate([c.reshape(-1, 3), d.reshape(-1, 3)])          # (a, c, b) and (a, d, c): normals up
    p3 = np.concatenate([b.reshape(-1, 3), c.reshape(-1, 3)])
    t = len(p1)
    e1, e2 = p2 - p1, p3 - p1
    nrm = np.cross(e2, e1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    g = np.zeros((t, 12)); g[:, 0:3], g[:, 3:6], g[:, 6:9], g[:, 9] = p1, e1, e2, cs.arrays["tri_geom"][0, 9]
    nn = np.zeros((t, 12)); nn[:, 0:3] = nn[:, 3:6] = nn[:, 6:9] = nrm
    A = cs.arrays
    A["tri_geom"], A["tri_norm"] = g, nn
    A["tri_leaf"] = np.arange(t, dtype=np.int32)
    for k, v in (("leaf_kind", A["leaf_kind"][0]), ("leaf_xform", A["leaf_xform"][0]), ("leaf_material", A["leaf_material"][0]), ("leaf_closed", 0)):
        A[k] = np.full(t, v, np.int32)
    A["leaf_tri"] = np.arange(t, dtype=np.int32)
    A["leaf_minmax"] = np.zeros((t, 2))
    A["item_prim"] = ((np.uint32(K_TRIANGLE) << np.uint32(28)) | np.arange(t, dtype=np.uint32)).astype(np.uint32).view(np.int32)
    A["item_lo"] = np.minimum(np.minimum(p1, p2), p3)
    A["item_hi"] = np.maximum(np.maximum(p1, p2), p3)
    cs.counts.update(n_leaves=t, n_tris=t, n_items=t)
    cs.leaf_objects = [seed_tri] * t
    return cs


cam = Camera(1920, 1080, 0.9, view_transform(point(0.5, 3.0, -7.5), point(0, 0, 0), vector(0, 1, 0)))
for n in [int(a) for a in sys.argv[1:]] or [1000]:
    t0 = time.perf_counter()
    cs = height_field(n)
    t1 = time.perf_counter()
    with _ffi.Scene(cs, build_on_device=True) as sc:
        t2 = time.perf_counter()
        st = {}
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
        for _ in range(8):
            sc.render(cam, precision="fp64", stats=st)
        st32 = {}
        for _ in range(8):
            f32, _ = sc.render(cam, precision="fp32", stats=st32)
        info = sc.info()
    hit = (ids >= 0).mean()
    print("N %d: %d triangles, arrays %.1f s, create with RTC_BUILD_ON_DEVICE %.2f s, depth %d, %.0f MB on the device; 1080p fp64 %.3f ms fp32 %.3f ms, "
          "%.0f %% of the pixels hit the mesh, %d rays" % (n, cs.counts["n_tris"], t1 - t0, t2 - t1, info["bvh_depth"], info["device_bytes_fp64"] / 1e6,
                                                        st["kernel_ms"], st32["kernel_ms"], 100 * hit, sum(st["rays"])), flush=True)
    assert hit > 0.3 and np.isfinite(rgb).all() and len(np.unique(ids)) > 10000
    if cs.counts["n_tris"] <= 2500000:
        t3 = time.perf_counter()
        with _ffi.Scene(cs) as sc:
            t4 = time.perf_counter()
            h, hi = sc.render(cam, precision="fp64", want_ids=True)
        d = (np.abs(h - rgb).max(axis=2) > 1e-9) | (hi != ids)
        print("      host-built tree: create %.2f s; %d of %d pixels differ between the two trees' frames (ties on shared edges)" % (t4 - t3, int(d.sum()), d.size), flush=True)
        assert d.mean() < 1e-3
