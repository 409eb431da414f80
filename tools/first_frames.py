"""Kernel ms of the first frames of a fresh view (development aid): first_frames.py scene W H [precision]
Frame 1 runs in plain tile order, frame 2 sorted by frame 1's costs, frame 3 on with the dearest units split."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
if os.environ.get("RTC_LIB"):
    _ffi.LIB_PATH = os.environ["RTC_LIB"]
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else "fp64"
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(w, h)
cs = compile_scene(world)
for trial in range(3):
    with _ffi.Scene(cs) as sc:
        ms, hs = [], set()
        for k in range(6):
            st = {}
            rgb, ids = sc.render(cam, precision=prec, want_ids=True, stats=st)
            ms.append(st["kernel_ms"])
            hs.add(hashlib.sha256(np.ascontiguousarray(rgb).tobytes()).hexdigest()[:8])
        print("%s %dx%d %s trial %d: frames 1..6 kernel ms %s  hashes %s" % (name, w, h, prec, trial, " ".join("%.3f" % m for m in ms), sorted(hs)), flush=True)
