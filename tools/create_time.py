import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ray_tracer_challenge_amd import _ffi, renderer
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cam, world, _ = load_world(os.path.join(ROOT, "scenes", "dragons.npz"))
cam = cam.resized(1920, 1080)
cs = compile_scene(world)
_ffi.Scene(cs).close()
for dev in (True, False, True):
    print("--- build_on_device", dev, flush=True)
    t0 = time.perf_counter(); sc = _ffi.Scene(cs, build_on_device=dev); t1 = time.perf_counter()
    sc.render(cam); t2 = time.perf_counter(); sc.close(); t3 = time.perf_counter()
    print("create %.1f ms, first render (host buffers) %.1f ms, close %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
import cProfile, pstats
t0 = time.perf_counter(); renderer.render_multi(cam, world); print("render_multi %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); renderer.render_multi(cam, world); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
