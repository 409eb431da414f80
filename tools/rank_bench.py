"""Kernel time of one rank's share of the frame at N = 1, 2, 4, 8 ranks (tile i -> rank i mod N), on one GPU:
what strong scaling can at best look like before the gather.  Usage: rank_bench.py [lib.so] [scene W H]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
from ray_tracer_challenge_amd.dist import rank_tiles
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
if args and args[0].endswith(".so"):
    _ffi.LIB_PATH = os.path.abspath(args.pop(0))
name, w, h = (args + ["dragons", "1920", "1080"])[:3]
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(int(w), int(h))
with _ffi.Scene(compile_scene(world)) as sc:
    for prec in ("fp64", "fp32"):
        row = []
        for n in (1, 2, 4, 8):
            worst = 0.0
            for r in range(n if n <= 2 else 2):          # two ranks are enough to see the spread
                tiles = rank_tiles(cam.hsize, cam.vsize, n, r)
                ms = []
                for _ in range(10):
                    st = {}
                    sc.render_tiles(cam, tiles, precision=prec, stats=st)
                    ms.append(st["kernel_ms"])
                worst = max(worst, min(ms[3:]))
            row.append("N=%d %.3f ms" % (n, worst))
        print(name, prec, " | ".join(row), flush=True)
