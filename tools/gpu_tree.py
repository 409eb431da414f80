"""Host SAH tree against the GPU-built triangle BVH, per builder setting: build time, depth, frame time (development aid).
Usage: gpu_tree.py [builder[:radius] ...]   e.g. gpu_tree.py lbvh ploc:8 ploc:16 ploc:32"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from ray_tracer_challenge_amd import _ffi
    from ray_tracer_challenge_amd.compile import compile_scene
    from ray_tracer_challenge_amd.scene_io import load_world
    for name in ("dragons", "teapot"):
        cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
        cam = cam.resized(1920, 1080)
        cs = compile_scene(world)
        with _ffi.Scene(cs) as sc:
            def frame(prec):
                ms = []
                for _ in range(10):
                    st = {}
                    sc.render(cam, precision=prec, stats=st)
                    ms.append(st["kernel_ms"])
                return min(ms[5:])
            host = (frame("fp64"), frame("fp32"), sc.info()["bvh_depth"])
            sc.build_bvh_on_device()
            t0 = time.perf_counter()
            for _ in range(5):
                sc.build_bvh_on_device()
            t_build = (time.perf_counter() - t0) / 5 * 1e3
            dev = (frame("fp64"), frame("fp32"), sc.info()["bvh_depth"])
        print("  %-8s host SAH: depth %d, frame fp64 %.3f fp32 %.3f ms | GPU build %.2f ms: depth %d, frame fp64 %.3f (x %.2f) fp32 %.3f (x %.2f)" % (
            name, host[2], host[0], host[1], t_build, dev[2], dev[0], dev[0] / host[0], dev[1], dev[1] / host[1]), flush=True)
    sys.exit(0)
for v in (sys.argv[1:] or ["lbvh", "ploc:16"]):
    env = dict(os.environ)
    b, _, r = v.partition(":")
    env["RTC_GPU_BUILDER"] = b
    if r:
        env["RTC_PLOC_RADIUS"] = r
    print("== %s" % v, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=True, timeout=300)
