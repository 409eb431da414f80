"""Where does the fp32 mode differ from fp64 by more than one PPM level? (development aid)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi  # noqa: E402
from ray_tracer_challenge_amd.canvas import quantize  # noqa: E402
from ray_tracer_challenge_amd.compile import compile_scene  # noqa: E402
from ray_tracer_challenge_amd.scene_io import load_world  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
size = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
if size:
    cam = cam.resized(*size)
with _ffi.Scene(compile_scene(world)) as sc:
    a, ia = sc.render(cam, precision="fp64", want_ids=True)
    for eps in (0.0, 1e-5, 5e-5, 1e-4, 2e-4, 5e-4, 1e-3):
        b, ib = sc.render(cam, precision="fp32", want_ids=True, shade_eps=eps)
        d = np.abs(quantize(a, True).astype(int) - quantize(b, True).astype(int)).max(axis=2).T
        bad = np.argwhere(d > 1)
        print("eps %g: %d bad of %d (%.4f%%), id mismatches %d" % (eps, len(bad), d.size, 100.0 * len(bad) / d.size, (ia != ib).sum()))
    b, ib = sc.render(cam, precision="fp32", want_ids=True)
    d = np.abs(quantize(a, True).astype(int) - quantize(b, True).astype(int)).max(axis=2).T
    bad = np.argwhere(d > 1)
    for x, y in bad[:40]:
        print(x, y, "ids", ia[x, y], ib[x, y], "kind", compile_scene(world).arrays["leaf_kind"][ia[x, y]] if ia[x, y] >= 0 else -1,
              "f64", np.round(a[x, y], 4), "f32", np.round(b[x, y], 4))
    # per primary-hit leaf: how many of its pixels are off
    for leaf in np.unique(ia):
        sel = ia == leaf
        print("leaf %d: %d px, %d off" % (leaf, sel.sum(), (d.T[sel] > 1).sum()))
