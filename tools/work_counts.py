"""Ray and work counters of one frame of a scene pack.  Usage: work_counts.py scene W H [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else "fp64"
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(w, h)
with _ffi.Scene(compile_scene(world)) as sc:
    st = {}
    sc.render(cam, precision=prec, stats=st, count_work=True)
    for _ in range(3):
        t = {}
        sc.render(cam, precision=prec, stats=t)
    names = ["bvh_node_visits", "triangle_tests", "analytic_leaf_tests", "csg_evals", "shaded_hits", "light_samples", "container_passes", "pattern_evals"]
    print(name, "%dx%d" % (w, h), prec, "kernel %.3f ms" % t["kernel_ms"])
    print("  rays primary/shadow/reflection/refraction", st["rays"], "shadow queries answered without a ray", st["shadow_rays_culled"])
    print("  " + ", ".join("%s %d" % (n, v) for n, v in zip(names, st["work"])))
