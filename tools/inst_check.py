import os, sys, hashlib, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for w, h in ((480, 192), (1920, 1080)):
    cam, world, _ = load_world(os.path.join(ROOT, "scenes", "dragons.npz"))
    cam = cam.resized(w, h)
    out = {}
    for inst in (False, True):
        cs = compile_scene(world, instancing=inst)
        with _ffi.Scene(cs) as sc:
            res = []
            for prec in ("fp64", "fp32"):
                rgb, ids = sc.render(cam, precision=prec, want_ids=True)
                ms = []
                for _ in range(14):
                    st = {}
                    sc.render(cam, precision=prec, stats=st)
                    ms.append(st["kernel_ms"])
                res.append((rgb, ids, sorted(ms[8:])[3], st["rays"], st["shadow_rays_culled"]))
            out[inst] = res
            print(w, h, "instancing", inst, "device bytes", sc.info()["device_bytes_fp64"], sc.info()["device_bytes_fp32"], "fp64 %.3f ms fp32 %.3f ms" % (res[0][2], res[1][2]), res[0][3], flush=True)
    for k, prec in enumerate(("fp64", "fp32")):
        a, b = out[False][k], out[True][k]
        print("  ", prec, "max |d rgb| %.3e" % np.abs(a[0] - b[0]).max(), "pixels differing > 1e-9:", int((np.abs(a[0] - b[0]).max(axis=2) > 1e-9).sum()), "ids differing:", int((a[1] != b[1]).sum()), flush=True)
