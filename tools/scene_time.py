"""Steady-state kernel time of one scene, several repetitions (development aid). Usage: scene_time.py scene W H [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(w, h)
cs = compile_scene(world)
for prec in ("fp64", "fp32"):
    out = []
    for _ in range(reps):
        with _ffi.Scene(cs) as sc:
            ms = []
            for _ in range(8):
                st = {}
                sc.render(cam, precision=prec, stats=st)
                ms.append(st["kernel_ms"])
            out.append(min(ms[3:]))
    print(name, prec, " ".join("%.3f" % v for v in out), flush=True)
