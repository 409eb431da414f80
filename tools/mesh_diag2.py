"""Is a pixel's difference from the oracle a rounding flip or systematic?  The kernel and the oracle under camera turns of
+-1e-9 ... 1e-6 rad.  Usage: mesh_diag2.py seed x y depth"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_fuzz as F
from oracle.oracle import Oracle
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.matrix import rotation_x, rotation_y
from ray_tracer_challenge_amd.renderer import Camera
seed, x, y, depth = [int(a) for a in sys.argv[1:5]]
cam, world, rng = F.random_mesh_world(seed)
orc = Oracle(world)
with _ffi.Scene(compile_scene(world), device=0) as sc:
    for ang in (0.0, 1e-9, -1e-9, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6):
        for rot, nm in ((rotation_x, "x"), (rotation_y, "y")):
            cam2 = Camera(cam.hsize, cam.vsize, cam.field_of_view, rot(ang) * cam.transform)
            k = sc.render(cam2, precision="fp64", max_depth=depth)[0][x, y]
            o = orc.render_pixels(cam2, np.array([x]), np.array([y]), remaining=depth)[0][0]
            print("turn %s %+.0e: kernel %s oracle %s |d| %.3g" % (nm, ang, k, o, np.abs(k - o).max()))
