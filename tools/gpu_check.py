"""Quick GPU-vs-oracle comparison on scene packs (development aid; the real checks are tests/)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import Oracle  # noqa: E402
from ray_tracer_challenge_amd import _ffi  # noqa: E402
from ray_tracer_challenge_amd.canvas import quantize  # noqa: E402
from ray_tracer_challenge_amd.compile import compile_scene  # noqa: E402
from ray_tracer_challenge_amd.scene_io import load_world  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(name, size=None, precision="fp64", stride=1):
    cam, world, objs = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
    if size:
        cam = cam.resized(*size)
    t0 = time.time()
    cs = compile_scene(world)
    t1 = time.time()
    st = {}
    with _ffi.Scene(cs) as sc:
        info = sc.info()
        rgb, ids = sc.render(cam, precision=precision, want_ids=True, stats=st)
        st2 = {}
        sc.render(cam, precision=precision, stats=st2)
    xs, ys = np.meshgrid(np.arange(0, cam.hsize, stride), np.arange(0, cam.vsize, stride), indexing="ij")
    t2 = time.time()
    o = Oracle(world)
    orgb, oleaf, ot, orays = o.render_pixels(cam, xs.ravel(), ys.ravel())
    t3 = time.time()
    g = rgb[xs.ravel(), ys.ravel()]
    gi = ids[xs.ravel(), ys.ravel()]
    d = np.abs(g - orgb)
    q = np.abs(quantize(g.reshape(1, -1, 3), True).astype(int) - quantize(orgb.reshape(1, -1, 3), True).astype(int))
    bad = (q.max(axis=2) > 1).sum()
    print("%-28s %s %4dx%-4d compile %.2fs bvh %d nodes depth %d | kernel %.3f ms (2nd %.3f) rays %s | oracle %.1fs rays %s | "
          "max|d| %.2e  >1/255: %d/%d  id mismatch %d" % (
              name, precision, cam.hsize, cam.vsize, t1 - t0, info["bvh_nodes"], info["bvh_depth"], st["kernel_ms"],
              st2["kernel_ms"], st["rays"], t3 - t2, list(orays), d.max(), bad, len(g), (gi != oleaf).sum()), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    prec = args[0] if args else "fp64"
    check("cover", precision=prec)
    check("csg_test", size=(256, 256), precision=prec)
    check("group", precision=prec)
    check("checkered_sphere", size=(200, 200), precision=prec)
    check("checkered_plane", size=(200, 200), precision=prec)
    check("checkered_cylinder", size=(200, 200), precision=prec)
    check("align_check_plane", size=(200, 200), precision=prec)
    check("checkered_cube", size=(400, 200), precision=prec)
    check("earth", size=(400, 200), precision=prec)
    check("skybox", size=(400, 200), precision=prec)
    check("shadow_glamour_shot_nojitter", size=(300, 120), precision=prec)
    check("teapot", precision=prec, stride=4)
    check("dragons", size=(480, 192), precision=prec, stride=24)
