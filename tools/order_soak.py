"""Launch-order feedback on random content: frames 2..10 of a view (sorted, split into parts with replicas, both fp64 register
budgets timed) must equal frame 1 (plain tile order) bit for bit -- colours, hit ids, ray counts -- on the fuzz generators'
worlds at a size where units do get split.  Usage: order_soak.py n_seeds [width height]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_fuzz as F
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd import compile as compile_mod
from ray_tracer_challenge_amd.compile import compile_scene
compile_mod.MIN_INSTANCE_TRIS = 32
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 400)
bad = split = 0
for seed in range(n):
    for gen, kw in ((F.random_mesh_world, {}), (F.random_instanced_world, {"instancing": True}), (F.random_world, {}), (F.random_csg_world, {})):
        cam, world = gen(seed)[:2]
        cam = cam.resized(w, h)
        with _ffi.Scene(compile_scene(world, **kw), device=0) as sc:
            for prec in ("fp64", "fp32"):
                first = None
                for k in range(10):
                    st = {}
                    rgb, ids = sc.render(cam, precision=prec, want_ids=True, stats=st)
                    got = (rgb.copy(), ids.copy(), tuple(st["rays"]), st["shadow_rays_culled"])
                    if first is None:
                        first = got
                    elif not (np.array_equal(got[0], first[0], equal_nan=True) and np.array_equal(got[1], first[1]) and got[2:] == first[2:]):
                        bad += 1
                        print("MISMATCH %s seed %d %s frame %d: %d px, rays %s vs %s" % (gen.__name__, seed, prec, k + 1,
                              int((np.abs(got[0] - first[0]).max(axis=2) > 0).sum()), got[2:], first[2:]), flush=True)
                li = sc.last_launch()
                split += int(li["n_entries"] > li["n_units"])
    if seed % 5 == 4:
        print("seed %d done: %d mismatches so far, %d views ended with split units" % (seed, bad, split), flush=True)
print("order soak: %d worlds x 2 precisions x 10 frames, %d mismatches, %d views with split units" % (4 * n, bad, split))
sys.exit(1 if bad else 0)
