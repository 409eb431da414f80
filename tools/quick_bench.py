"""Kernel-time table over a few scenes/precisions (development aid for A/B-ing kernel changes).
Usage: quick_bench.py [lib.so ...]  -- each library is timed in turn in one process."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi  # noqa: E402
from ray_tracer_challenge_amd.compile import compile_scene  # noqa: E402
from ray_tracer_challenge_amd.scene_io import load_world  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("dragons", (1920, 1080)), ("teapot", (1920, 1080)), ("cover", (1920, 1080)), ("shadow_glamour_shot_nojitter", (1200, 480)),
         ("csg_test", (1024, 1024)), ("earth", (1920, 1080))]
libs = sys.argv[1:] or [_ffi.LIB_PATH]
scenes = {}
for name, size in CASES:
    cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
    scenes[name] = (cam.resized(*size), compile_scene(world))
ref = {}
for lib in libs:
    _ffi._lib = None
    _ffi.LIB_PATH = os.path.abspath(lib)
    print("== %s" % lib, flush=True)
    for name, size in CASES:
        cam, cs = scenes[name]
        with _ffi.Scene(cs) as sc:
            row = []
            forms = [("fb", {}), ("nofb", {"feedback": False})]
            for prec, mega in [(p_, f_) for p_ in ("fp64", "fp32") for f_ in forms]:
                ms = []
                for it in range(6):
                    st = {}
                    rgb, _ = sc.render(cam, precision=prec, stats=st, **mega[1])
                    ms.append(st["kernel_ms"])
                mega = mega[0]
                key = (name, prec, mega)
                same = ""
                if key in ref:
                    same = " same" if np.array_equal(ref[key], rgb) else " DIFF(max %.2e)" % np.abs(ref[key] - rgb).max()
                else:
                    ref[key] = rgb
                rays = sum(st["rays"])
                if mega == "fb":
                    base = rgb
                else:
                    same += " =fb" if np.array_equal(base, rgb) else " !=fb(%d px)" % int((np.abs(base - rgb).max(axis=2) > 1e-9).sum())
                row.append("%s/%s %6.3f ms%s" % (prec, mega, min(ms[1:]), same))
        print("  %-30s %s" % (name, " | ".join(row)), flush=True)
