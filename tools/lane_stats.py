"""SIMD lane utilisation per kind of work: lane-level counts from the product library, wave-level counts from a
-DRTC_LANE_STATS build (every counter then counts wave-level executions x 65536).  Usage: lane_stats.py lanes.so [base.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["bvh_node_visits", "triangle_tests", "analytic_leaf_tests", "csg_evals", "shaded_hits", "light_samples", "container_passes", "pattern_evals", "wide_node_visits"]
prod = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else _ffi.LIB_PATH
res = {}
for lib in (prod, os.path.abspath(sys.argv[1])):
    _ffi._lib = None
    _ffi.LIB_PATH = lib
    for name, (w, h) in (("dragons", (1920, 1080)), ("teapot", (1920, 1080)), ("cover", (1920, 1080))):
        cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
        cam = cam.resized(w, h)
        with _ffi.Scene(compile_scene(world)) as sc:
            st = {}
            sc.render(cam, precision="fp64", stats=st, count_work=True, feedback=False)
            res[(lib == prod, name)] = list(st["work"]) + [st.get("wide_node_visits", 0)]
for name in ("dragons", "teapot", "cover"):
    lane, wave = res[(True, name)], res[(False, name)]
    print(name)
    for k, label in enumerate(NAMES):
        if lane[k]:
            wv = wave[k] / 65536.0
            print("   %-20s lane-level %11d  wave-level %10.0f  active lanes %.1f / 64" % (label, lane[k], wv, lane[k] / max(wv, 1)))
