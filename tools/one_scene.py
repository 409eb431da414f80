"""Render one scene pack N times (for rocprofv3 runs). Usage: one_scene.py name W H precision [n]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi  # noqa: E402
from ray_tracer_challenge_amd.compile import compile_scene  # noqa: E402
from ray_tracer_challenge_amd.scene_io import load_world  # noqa: E402
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, w, h, prec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 6
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(w, h)
with _ffi.Scene(compile_scene(world)) as sc:
    for _ in range(n):
        st = {}
        sc.render(cam, precision=prec, stats=st, feedback=not os.environ.get("RTC_NOFB"))
    print(st)
