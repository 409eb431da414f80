"""Host-built (binned SAH) vs GPU-built (LBVH) triangle BVH: build time and frame time.  Usage: bvh_bench.py [scene ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in (sys.argv[1:] or ["dragons", "teapot"]):
    cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
    cam = cam.resized(1920, 1080)
    cs = compile_scene(world)
    _ffi.Scene(cs).close()                                   # first touch of the device
    t0 = time.perf_counter()
    sc = _ffi.Scene(cs)
    t_create = time.perf_counter() - t0
    def frame(prec):
        ms = []
        for _ in range(6):
            st = {}
            sc.render(cam, precision=prec, stats=st)
            ms.append(st["kernel_ms"])
        return min(ms[2:])
    host = (frame("fp64"), frame("fp32"), sc.info())
    # refit of the host tree after a smooth deformation (amplitude as a fraction of the mesh extent)
    g0 = np.asarray(cs.arrays["tri_geom"], np.float64).reshape(-1, 12)
    nrm = cs.arrays["tri_norm"]
    ext = float((g0[:, 0:3].max(axis=0) - g0[:, 0:3].min(axis=0)).max())
    refits = []
    for amp in (0.0, 0.01, 0.05, 0.2):
        g = g0.copy()
        g[:, 0:3] += amp * ext * np.sin((6.0 / ext) * g0[:, [1, 2, 0]])
        sc.update_triangles(g, nrm, refit=True)
        t0 = time.perf_counter()
        sc.update_triangles(g, nrm, refit=True)
        t_refit = time.perf_counter() - t0
        f_refit = frame("fp64")
        import torch
        tg = torch.from_numpy(np.ascontiguousarray(g)).to("cuda:0"); tn = torch.from_numpy(np.ascontiguousarray(nrm, dtype=np.float64)).to("cuda:0")
        torch.cuda.synchronize()
        sc.update_triangles_device(tg.data_ptr(), tn.data_ptr(), len(g), refit=True)
        t0 = time.perf_counter()
        sc.update_triangles_device(tg.data_ptr(), tn.data_ptr(), len(g), refit=True)
        t_dev = (time.perf_counter() - t0) * 1e3
        sc.update_triangles_device(tg.data_ptr(), tn.data_ptr(), len(g), refit=False)
        t0 = time.perf_counter()
        sc.update_triangles_device(tg.data_ptr(), tn.data_ptr(), len(g), refit=False)
        t_dev_b = (time.perf_counter() - t0) * 1e3
        sc.close()
        sc = _ffi.Scene(cs)
        sc.update_triangles(g, nrm, refit=True)
        sc.update_triangles(g, nrm)
        t0 = time.perf_counter()
        sc.update_triangles(g, nrm)
        t_upd = time.perf_counter() - t0
        refits.append((amp, t_refit * 1e3, f_refit, t_upd * 1e3, frame("fp64"), t_dev, t_dev_b))
        sc.close()
        sc = _ffi.Scene(cs)                                  # back to the host SAH tree
    sc.build_bvh_on_device()                                 # warm (allocations, code load)
    t0 = time.perf_counter()
    for _ in range(5):
        sc.build_bvh_on_device()
    t_build = (time.perf_counter() - t0) / 5
    dev = (frame("fp64"), frame("fp32"), sc.info())
    sc.close()
    print("%-10s %d triangles | rtc_scene_create (host SAH build + upload) %.1f ms | rtc_scene_build_bvh_device %.2f ms" % (
        name, cs.counts["n_tris"], t_create * 1e3, t_build * 1e3))
    print("           host tree: depth %d, %d nodes, frame fp64 %.3f ms fp32 %.3f ms | GPU tree: depth %d, %d nodes, frame fp64 %.3f ms fp32 %.3f ms" % (
        host[2]["bvh_depth"], host[2]["bvh_nodes"], host[0], host[1], dev[2]["bvh_depth"], dev[2]["bvh_nodes"], dev[0], dev[1]), flush=True)
    for amp, t_refit, f_refit, t_upd, f_upd, t_dev, t_dev_b in refits:
        print("           deformed by %4.0f %% of the extent: upload + refit %.2f ms (rows on the device: %.2f) -> frame fp64 %.3f ms | upload + rebuild %.2f ms (%.2f) -> frame %.3f ms" % (
            100 * amp, t_refit, t_dev, f_refit, t_upd, t_dev_b, f_upd), flush=True)
