"""Where does a tests/test_gpu_fuzz.py mesh world's frame part from the oracle's?  Usage: mesh_diag.py seed [seed ...]
For every pixel that explain_mismatches leaves unexplained: ids, colours, the oracle's hits along the camera ray, and the
depth at which the two frames part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_fuzz as F
from conftest import explain_mismatches
from oracle.oracle import Oracle
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene

for seed in [int(a) for a in sys.argv[1:]]:
    cam, world, rng = F.random_mesh_world(seed)
    w, h = cam.hsize, cam.vsize
    xs, ys = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    orc = Oracle(world)
    want, leaf, t_hit, _ = orc.render_pixels(cam, xs.ravel(), ys.ravel())
    want, leaf = want.reshape(w, h, 3), leaf.reshape(w, h)
    cs = compile_scene(world)
    with _ffi.Scene(cs, device=0) as sc:
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
        un, classes = explain_mismatches(orc, cam, xs.ravel(), ys.ravel(), rgb.reshape(-1, 3), want.reshape(-1, 3), ids.ravel(), leaf.ravel())
        print("seed %d: classes %s unexplained %s; lights %s" % (seed, classes, un, [type(l).__name__ for l in world.lights]))
        by_depth = {d: sc.render(cam, precision="fp64", want_ids=True, max_depth=d)[0] for d in range(6)}
        f32, i32 = sc.render(cam, precision="fp32", want_ids=True)
    for (x, y) in un:
        o, d = orc.ray_for_pixel(cam, x, y)
        t, node = orc.intersect_world(o, d, cap=4096)
        leaf_of = np.asarray(orc.ex.a["leaf_id"])
        pos = np.flatnonzero(t > 0)[:4]
        print("  px (%d,%d): kernel id %d (fp32 %d) oracle leaf %d | kernel %s oracle %s" % (x, y, ids[x, y], i32[x, y], leaf[x, y], rgb[x, y], want[x, y]))
        print("     oracle hits (t, leaf): %s" % [(float(t[k]), int(leaf_of[node[k]])) for k in pos])
        for dep in range(6):
            wd = orc.render_pixels(cam, np.array([x]), np.array([y]), remaining=dep)[0][0]
            print("     depth %d: kernel %s oracle %s  |d| %.3g" % (dep, by_depth[dep][x, y], wd, np.abs(by_depth[dep][x, y] - wd).max()))
        for k in (leaf[x, y], ids[x, y]):
            if k >= 0:
                ob = cs.leaf_objects[k]; m = ob.material
                print("     leaf %d: %s refl %.2f transp %.2f ri %.3f pattern %s" % (k, type(ob).__name__, m.reflective, m.transparency, m.refractive_index, type(m.pattern).__name__ if m.pattern is not None else "-"))
