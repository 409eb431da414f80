"""Kernel time of rank 0's share at N ranks for several bounce depths (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
from ray_tracer_challenge_amd.dist import rank_tiles
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "dragons"
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(1920, 1080)
with _ffi.Scene(compile_scene(world)) as sc:
    for n in (1, 8):
        tiles = rank_tiles(cam.hsize, cam.vsize, n, 0)
        row = []
        for depth in (0, 1, 2, 3, 5):
            ms = []
            for _ in range(6):
                st = {}
                sc.render_tiles(cam, tiles, precision="fp64", max_depth=depth, stats=st)
                ms.append(st["kernel_ms"])
            row.append("depth %d %.3f ms (%d rays)" % (depth, min(ms[2:]), sum(st["rays"])))
        print(name, "N=%d" % n, " | ".join(row), flush=True)
