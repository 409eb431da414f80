"""Ray and work counters of single tiles of a frame (development aid): what the dearest blocks of a launch are made of.
Usage: tile_work.py scene W H x,y [x,y ...]   (pixel coordinates; the 8x8-aligned block around each is rendered alone)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(w, h)
names = ["node_visits", "tri_tests", "leaf_tests", "csg", "shaded", "light_samples", "container", "pattern"]
with _ffi.Scene(compile_scene(world)) as sc:
    for arg in sys.argv[4:]:
        x, y = [int(v) for v in arg.split(",")]
        tx = _ffi.tiles_for(w, h, 8)[0]
        tile = (y // 8) * tx + x // 8
        for depth in (5, 0):
            st = {}
            rgb, ids = sc.render_tiles(cam, np.array([tile], np.int32), precision="fp64", tile_size=8, stats=st, count_work=True, want_ids=True, max_depth=depth, feedback=False)
            t = {}
            sc.render_tiles(cam, np.array([tile], np.int32), precision="fp64", tile_size=8, stats=t, max_depth=depth, feedback=False)
            print("block at (%d,%d) depth %d: %.1f us alone; rays p/s/refl/refr %s culled %d; %s; distinct leaves %d" % (
                x, y, depth, 1000 * t["kernel_ms"], st["rays"], st["shadow_rays_culled"], ", ".join("%s %d" % (n, v) for n, v in zip(names, st["work"]) if v), len(np.unique(ids))))
