"""How the launch-order split thresholds move the frame time (development aid).  The order is fixed from the costs of a view's
second frame, so one scene handle = one sample of the order; each setting is sampled with several handles.
Usage: split_sweep.py scene W H precision "s4,s16" ["s4,s16" ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from ray_tracer_challenge_amd import _ffi
    from ray_tracer_challenge_amd.compile import compile_scene
    from ray_tracer_challenge_amd.scene_io import load_world
    name, w, h, prec = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
    cam = cam.resized(w, h)
    cs = compile_scene(world)
    out = []
    for rep in range(6):
        with _ffi.Scene(cs) as sc:
            ms = []
            for _ in range(12):
                st = {}
                sc.render(cam, precision=prec, stats=st)
                ms.append(st["kernel_ms"])
            out.append(float(np.median(ms[3:])))
    print("   handles: %s  -> median %.3f  min %.3f  max %.3f" % (" ".join("%.3f" % v for v in out), np.median(out), min(out), max(out)), flush=True)
    sys.exit(0)
name, w, h, prec = sys.argv[1:5]
for setting in sys.argv[5:]:
    s4, s16 = setting.split(",")
    env = dict(os.environ, RTC_SPLIT4=s4, RTC_SPLIT16=s16)
    print("%s %sx%s %s split4=%s split16=%s" % (name, w, h, prec, s4, s16), flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, w, h, prec], env=env, check=True, timeout=300)
