"""Frame time against the forced split target RTC_SPLIT_T (fraction of the dearest unit's cost; development aid).
Usage: t_sweep.py frac [frac ...]     (0 = the library's own rule)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("dragons", 1920, 1080, "fp64", 1), ("dragons", 1920, 1080, "fp32", 1), ("dragons", 480, 192, "fp64", 1), ("dragons", 1920, 1080, "fp64", 8),
         ("cover", 1920, 1080, "fp64", 1), ("cover", 1920, 1080, "fp32", 1), ("teapot", 1920, 1080, "fp64", 1), ("teapot", 1920, 1080, "fp64", 8)]
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from ray_tracer_challenge_amd import _ffi
    from ray_tracer_challenge_amd.compile import compile_scene
    from ray_tracer_challenge_amd.scene_io import load_world
    from ray_tracer_challenge_amd.dist import rank_tiles
    packs = {}
    row = []
    for name, w, h, prec, ranks in CASES:
        if name not in packs:
            cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
            packs[name] = (cam, compile_scene(world))
        cam, cs = packs[name]
        cam = cam.resized(w, h)
        out = []
        for rep in range(3):
            with _ffi.Scene(cs) as sc:
                ms = []
                for _ in range(10):
                    st = {}
                    if ranks == 1:
                        sc.render(cam, precision=prec, stats=st)
                    else:
                        sc.render_tiles(cam, rank_tiles(cam.hsize, cam.vsize, ranks, 0), precision=prec, stats=st)
                    ms.append(st["kernel_ms"])
                out.append(float(np.median(ms[3:])))
        row.append("%.3f" % np.median(out))
    print(" ".join("%7s" % v for v in row), flush=True)
    sys.exit(0)
print("frac   " + " ".join("%s%s@%d/%d" % (n[:3], p[2:], w, r) for n, w, h, p, r in CASES), flush=True)
for frac in sys.argv[1:]:
    print("%-6s" % frac, end=" ", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, RTC_SPLIT_T=frac), check=True, timeout=300,
                   stderr=subprocess.DEVNULL)
