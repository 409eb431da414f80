"""SHA-256 of the frames (fp64 and fp32), hit-id maps and ray counts of every parity scene, plus steady-state kernel
times (development aid).  Run once with a known-good library to record a baseline, again after a kernel change to
compare: `frame_hashes.py out.json [baseline.json]`."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_challenge_amd import _ffi  # noqa: E402
from ray_tracer_challenge_amd.compile import compile_scene  # noqa: E402
from ray_tracer_challenge_amd.scene_io import load_world  # noqa: E402

CASES = [("cover", None), ("csg_test", (192, 192)), ("group", None), ("checkered_sphere", (160, 160)),
         ("checkered_plane", (160, 160)), ("checkered_cylinder", (160, 160)), ("align_check_plane", (160, 160)),
         ("checkered_cube", (320, 160)), ("earth", (320, 160)), ("skybox", (320, 160)),
         ("shadow_glamour_shot_nojitter", (240, 96)), ("shadow_glamour_shot", (240, 96)), ("teapot", None),
         ("dragons", (480, 192)), ("perturbed", (320, 200)), ("patterns", None), ("uvmaps", None),
         ("dragons", (1920, 1080)), ("cover", (1920, 1080)), ("teapot", (1920, 1080)), ("csg_test", (1024, 1024)),
         ("shadow_glamour_shot", (1200, 480)), ("earth", (1920, 1080))]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out_path = sys.argv[1]
    base = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
    out = {}
    scenes = {}
    for name, size in CASES:
        if name not in scenes:
            cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
            scenes[name] = (cam, _ffi.Scene(compile_scene(world)))
        cam, sc = scenes[name]
        if size:
            cam = cam.resized(*size)
        key = "%s@%dx%d" % (name, cam.hsize, cam.vsize)
        row = {}
        for prec in ("fp64", "fp32"):
            st = {}
            rgb, ids = sc.render(cam, precision=prec, want_ids=True, stats=st, seed=1)
            ms = []
            for _ in range(5):
                s2 = {}
                sc.render(cam, precision=prec, stats=s2, seed=1)
                ms.append(s2["kernel_ms"])
            row[prec] = {"rgb": sha(rgb), "ids": sha(ids), "rays": st["rays"], "culled": st["shadow_rays_culled"],
                         "kernel_ms": round(min(ms[2:]), 4)}
        out[key] = row
        line = "%-42s fp64 %.3f ms  fp32 %.3f ms" % (key, row["fp64"]["kernel_ms"], row["fp32"]["kernel_ms"])
        if base and key in base:
            for prec in ("fp64", "fp32"):
                b, r = base[key][prec], row[prec]
                same = [k for k in ("rgb", "ids", "rays", "culled") if b[k] == r[k]]
                line += "  | %s: %s (was %.3f ms)" % (prec, "IDENTICAL" if len(same) == 4 else "same " + ",".join(same), b["kernel_ms"])
        print(line, flush=True)
        with open(out_path, "w") as f:
            json.dump(out, f, indent=1)
    for _, sc in scenes.values():
        sc.close()


if __name__ == "__main__":
    main()
