"""Where a caller's wall time goes for one frame through the host-buffer boundary.  Usage: e2e_time.py [scene] [w h]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
from ray_tracer_challenge_amd import renderer
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "dragons"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(w, h)
def clock(f, n=1):
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
t_compile, cs = clock(lambda: compile_scene(world))
_ffi.Scene(cs).close()
t_create_dev, sc = clock(lambda: _ffi.Scene(cs, build_on_device=True))
sc.render(cam, precision="fp64")
t_first_dev, _ = clock(lambda: sc.render(cam, precision="fp64", feedback=False), 5)
sc.close()
t_create, sc = clock(lambda: _ffi.Scene(cs))
st = {}
sc.render(cam, precision="fp64", stats=st)
t_render, _ = clock(lambda: sc.render(cam, precision="fp64", stats=st), 5)
t_ids, _ = clock(lambda: sc.render(cam, precision="fp64", want_ids=True), 5)
t_ppm, _ = clock(lambda: sc.render_ppm(cam, precision="fp64"), 5)
sc.close()
t_api, _ = clock(lambda: renderer.render_multi(cam, world))
t_api2, _ = clock(lambda: renderer.render_multi(cam, world))
print("%s %dx%d: rtc_scene_create with RTC_BUILD_ON_DEVICE %.1f ms, frame in plain order on that tree %.2f ms" % (name, w, h, t_create_dev, t_first_dev))
print("%s %dx%d: compile_scene %.1f ms | rtc_scene_create %.1f ms | rtc_render (host buffers) %.2f ms of which kernel %.2f ms | with hit ids %.2f ms | rtc_render_ppm %.2f ms | render_multi(cam, world) %.1f ms, again %.1f ms"
      % (name, w, h, t_compile, t_create, t_render, st["kernel_ms"], t_ids, t_ppm, t_api, t_api2))
