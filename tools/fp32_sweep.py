"""fp32 shade_eps sweep over all scene packs against the fp64 HIP render (development aid)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi  # noqa: E402
from ray_tracer_challenge_amd.canvas import quantize  # noqa: E402
from ray_tracer_challenge_amd.compile import compile_scene  # noqa: E402
from ray_tracer_challenge_amd.scene_io import load_world  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("cover", None), ("cover", (400, 400)), ("csg_test", (256, 256)), ("group", None), ("group", (900, 300)), ("checkered_sphere", (200, 200)),
         ("checkered_plane", (200, 200)), ("checkered_cylinder", (200, 200)), ("align_check_plane", (200, 200)),
         ("checkered_cube", (400, 200)), ("earth", (400, 200)), ("skybox", (400, 200)),
         ("shadow_glamour_shot_nojitter", (300, 120)), ("teapot", None), ("teapot", (512, 512)), ("dragons", (480, 192)), ("dragons", (1920, 1080))]
EPS = [0.0]
print("%-30s %-10s " % ("scene", "size") + " ".join("%9g" % e for e in EPS))
for name, size in CASES:
    cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
    if size:
        cam = cam.resized(*size)
    with _ffi.Scene(compile_scene(world)) as sc:
        a, ia = sc.render(cam, precision="fp64", want_ids=True)
        qa = quantize(a, True).astype(int)
        row = []
        for eps in EPS:
            b, ib = sc.render(cam, precision="fp32", want_ids=True, shade_eps=eps)
            d = np.abs(qa - quantize(b, True).astype(int)).max(axis=2)
            row.append("%5.3f/%4.2f" % (100.0 * (d > 1).mean(), 100.0 * (ia != ib).mean()))
    print("%-30s %-10s " % (name, "%dx%d" % (cam.hsize, cam.vsize)) + " ".join(row), flush=True)
print("cells: %% pixels off by more than one PPM level / %% hit-id mismatches, fp32 vs fp64")
