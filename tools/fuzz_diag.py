"""Which objects do the fp32/fp64 differences of a tests/test_gpu_fuzz.py world sit on?  Usage: fuzz_diag.py seed [seed ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_fuzz as F
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.canvas import quantize
from ray_tracer_challenge_amd.compile import compile_scene

def describe(p, d=0):
    if p is None:
        return "-"
    n = type(p).__name__
    subs = [getattr(p, k) for k in ("p1", "p2", "p3", "uv_pattern") if hasattr(p, k)] + list(getattr(p, "faces", []))
    extra = " f=%.2f s=%.2f o=%d" % (p.freq, p.scale_factor, p.octaves) if n == "PerturbedPattern" else ""
    return n + extra + ("(" + ", ".join(describe(q, d + 1) for q in subs) + ")" if subs else "")

for seed in [int(a) for a in sys.argv[1:]]:
    cam, world = (F.random_texture_world(seed - 100000) if seed >= 100000 else F.random_world(seed))
    cs = compile_scene(world)
    with _ffi.Scene(cs) as sc:
        a, ia = sc.render(cam, precision="fp64", want_ids=True)
        b, ib = sc.render(cam, precision="fp32", want_ids=True)
    d = np.abs(quantize(a, True).astype(int) - quantize(b, True).astype(int)).max(axis=2).T > 1     # [x][y]
    print("seed %d: %d px off of %d, lights %s" % (seed, d.sum(), d.size, [type(l).__name__ for l in world.lights]))
    for leaf in np.unique(ia):
        sel = ia == leaf
        if d[sel].sum() == 0:
            continue
        o = cs.leaf_objects[leaf] if leaf >= 0 else None
        m = o.material if o is not None else None
        print("   leaf %3d %-14s px %4d off %3d | refl %.2f transp %.2f ri %.3f | %s" % (
            leaf, type(o).__name__, sel.sum(), d[sel].sum(), m.reflective if m else 0, m.transparency if m else 0, m.refractive_index if m else 0,
            describe(m.pattern) if m else ""))
    if os.environ.get("EPS_SWEEP"):
        with _ffi.Scene(cs) as sc:
            for depth in (0, 1, 2, 5):
                a, _ = sc.render(cam, precision="fp64", max_depth=depth)
                row = []
                for eps in (0.0, 1e-5, 1e-4, 1e-3):
                    b, _ = sc.render(cam, precision="fp32", max_depth=depth, shade_eps=eps)
                    row.append("eps %g: %d" % (eps, (np.abs(quantize(a, True).astype(int) - quantize(b, True).astype(int)).max(axis=2) > 1).sum()))
                a2, _ = sc.render(cam, precision="fp64", max_depth=depth, shade_eps=1e-5)
                row.append("fp64 eps 1e-5 vs 1e-6: %d" % (np.abs(quantize(a, True).astype(int) - quantize(a2, True).astype(int)).max(axis=2) > 1).sum())
                print("   depth %d: %s" % (depth, " | ".join(row)))
