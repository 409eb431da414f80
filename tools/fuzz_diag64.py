"""Where does fp64 differ from the oracle on a tests/test_gpu_fuzz.py world?  Usage: fuzz_diag64.py seed [seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_fuzz as F
from oracle.oracle import Oracle
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
for seed in [int(a) for a in sys.argv[1:]]:
    cam, world = F.random_world(seed)
    xs, ys = np.meshgrid(np.arange(F.HSIZE), np.arange(F.VSIZE), indexing="ij")
    want, leaf, t, rays = Oracle(world).render_pixels(cam, xs.ravel(), ys.ravel())
    want, leaf, t = want.reshape(F.HSIZE, F.VSIZE, 3), leaf.reshape(F.HSIZE, F.VSIZE), t.reshape(F.HSIZE, F.VSIZE)
    cs = compile_scene(world)
    with _ffi.Scene(cs) as sc:
        st = {}
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
        rows = []
        for depth in range(0, 6):
            a, _ = sc.render(cam, precision="fp64", max_depth=depth)
            w2, _, _, _ = Oracle(world).render_pixels(cam, xs.ravel(), ys.ravel(), remaining=depth)
            rows.append(int((np.abs(a - w2.reshape(a.shape)).max(axis=2) > 1e-9).sum()))
    d = np.abs(rgb - want).max(axis=2)
    print("seed %d: %d px off, ids off %d, rays gpu %s culled %d oracle %s | off by depth 0..5: %s" % (
        seed, (d > 1e-9).sum(), (ids != leaf).sum(), st["rays"], st["shadow_rays_culled"], [int(r) for r in rays], rows))
    for l in np.unique(leaf[d > 1e-9]):
        o = cs.leaf_objects[l] if l >= 0 else None
        m = o.material if o is not None else None
        sel = (leaf == l) & (d > 1e-9)
        print("   primary leaf %3d %-14s off %3d max|d| %.3g | refl %.2f transp %.2f ri %.3f shadow %s | parent %s" % (
            l, type(o).__name__, sel.sum(), d[sel].max(), m.reflective, m.transparency, m.refractive_index, m.casts_shadow, type(getattr(o, "parent", None)).__name__))
    if os.environ.get("DETAIL"):
        for o in world.contains:
            for c in getattr(o, "children", [o]):
                if type(c).__name__ in ("Cone", "Cylinder"):
                    print("   ", type(c).__name__, "min %.4f max %.4f closed %s" % (c.minimum, c.maximum, c.closed), "in", type(o).__name__)
        bad = np.argwhere(d > 1e-9)
        for x, y in bad[:14]:
            print("    px (%d,%d): oracle leaf %d t %.9g rgb %s | gpu leaf %d rgb %s" % (x, y, leaf[x, y], t[x, y], np.round(want[x, y], 5), ids[x, y], np.round(rgb[x, y], 5)))
