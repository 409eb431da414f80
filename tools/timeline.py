"""Wave timeline of one frame from a -DRTC_TIMELINE build (development aid).
Usage: timeline.py lib.so scene W H [frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.scene_io import load_world
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
name, w, h = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
cam, world, _ = load_world(os.path.join(ROOT, "scenes", name + ".npz"))
cam = cam.resized(w, h)
with _ffi.Scene(compile_scene(world)) as sc:
    for _ in range(int(sys.argv[5]) if len(sys.argv) > 5 else 3):
        st = {}
        rgb, _ = sc.render(cam, precision="fp64", stats=st)
print("kernel_ms", st["kernel_ms"])
W8, H8 = w // 8, h // 8
s = rgb[:W8 * 8, :H8 * 8, 0].reshape(W8, 8, H8, 8).min(axis=(1, 3))
e = rgb[:W8 * 8, :H8 * 8, 1].reshape(W8, 8, H8, 8).max(axis=(1, 3))
cu = rgb[:W8 * 8, :H8 * 8, 2].reshape(W8, 8, H8, 8)[:, 0, :, 0]
t0 = s.min()
s, e = (s - t0) / 100.0, (e - t0) / 100.0       # microseconds
dur = e - s
span = e.max()
print("waves %d  span %.1f us  sum(dur) %.0f us  avg resident waves %.0f" % (dur.size, span, dur.sum(), dur.sum() / span))
print("wave duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
edges = np.linspace(0, span, 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    res = int(((s <= mid) & (e > mid)).sum())
    print("t=%7.1f us resident %5d  started %5d  finished %5d" % (mid, res, int(((s >= a) & (s < b)).sum()), int(((e >= a) & (e < b)).sum())))
print("last wave to start: %.1f us; waves still running at 50/75/90%% of span: %d %d %d" % (
    s.max(), ((e > 0.5 * span)).sum(), (e > 0.75 * span).sum(), (e > 0.9 * span).sum()))
ids = cu.astype(np.int64)
print("distinct (xcc,hw_id) %d" % len(np.unique(ids)))
# ten longest waves: where are they
flat = np.argsort(dur.ravel())[::-1][:10]
for f in flat:
    bx, by = np.unravel_index(f, dur.shape)
    print("  block (%d,%d) px (%d,%d) start %.1f dur %.1f" % (bx, by, bx * 8, by * 8, s[bx, by], dur[bx, by]))
