"""SIMD lane utilisation of the BVH node loop from a -DRTC_LANE_STATS build (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
for name, (w, h) in (("dragons", (1920, 1080)), ("teapot", (1920, 1080)), ("cover", (1920, 1080)), ("shadow_glamour_shot_nojitter", (1200, 480))):
    cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
    cam = cam.resized(w, h)
    with _ffi.Scene(compile_scene(world)) as sc:
        for form in ({}, {"feedback": False}):
            st = {}
            sc.render(cam, precision="fp64", stats=st, count_work=True, **form)
            node, wave = st["work"][0], st["work"][7] / 65536.0
            print("%-30s %-8s node visits %10d  wave-level iterations %10.0f  active lanes per iteration %.1f / 64" % (
                name, "plain" if form else "feedback", node, wave, node / max(wave, 1)))
