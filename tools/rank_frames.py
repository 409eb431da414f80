"""Kernel ms of successive frames of rank 0's share at N ranks (development aid): rank_frames.py scene W H prec N [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
from ray_tracer_challenge_amd.dist import rank_tiles
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, w, h, prec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(w, h)
cs = compile_scene(world)
for n in [int(a) for a in sys.argv[5:]]:
    with _ffi.Scene(cs) as sc:
        tiles = rank_tiles(cam.hsize, cam.vsize, n, 0)
        ms = []
        for _ in range(12):
            st = {}
            if n == 1:
                sc.render(cam, precision=prec, stats=st)
            else:
                sc.render_tiles(cam, tiles, precision=prec, stats=st)
            ms.append(st["kernel_ms"])
        print("N=%d: %s" % (n, " ".join("%.3f" % v for v in ms)), flush=True)
