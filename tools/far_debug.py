"""Debug aid: which pixels of the far-camera test differ from the oracle and by how much."""
import sys
import numpy as np
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_gpu_fuzz as T
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.renderer import Camera, World, point_light, view_transform
from ray_tracer_challenge_amd.matrix import translation, scaling, rotation_y
from ray_tracer_challenge_amd.vector import point, vector
from ray_tracer_challenge_amd import shapes as S
from oracle.oracle import Oracle

distance = float(sys.argv[1]) if len(sys.argv) > 1 else 1e5
rng = np.random.default_rng(77)
T._SPREAD[0] = 0.8
objs = [T._mesh(rng, 9, 14, smooth=True), S.Sphere(T._material(rng), translation(1.4, 0.2, 0.3) * scaling(0.4, 0.4, 0.4)),
        S.Cube(T._material(rng), translation(-1.3, -0.3, 0.2) * rotation_y(0.5) * scaling(0.35, 0.35, 0.35)),
        S.Cylinder(T._material(rng), translation(0.2, -0.9, -0.8) * scaling(0.3, 0.3, 0.3), minimum=-1.0, maximum=1.0, closed=True)]
for o in objs[1:]:
    o.material.pattern = None
lights = [point_light(point(-4, 6, -5), T._rgb(rng, 0.6, 1.0))]
eye = np.array([0.48, 0.31, -0.82]) / np.linalg.norm([0.48, 0.31, -0.82]) * distance
fov = 2 * np.arctan(2.4 / distance)
cam = Camera(96, 64, fov, view_transform(point(*eye), point(0.05, -0.1, 0), vector(0, 1, 0)))
world = World(lights, objs)
xs, ys = np.meshgrid(np.arange(cam.hsize), np.arange(cam.vsize), indexing="ij")
orc = Oracle(world)
want, leaf, _, rays = orc.render_pixels(cam, xs.ravel(), ys.ravel())
want, leaf = want.reshape(cam.hsize, cam.vsize, 3), leaf.reshape(cam.hsize, cam.vsize)
with _ffi.Scene(compile_scene(world), device=0) as sc:
    rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
d = np.abs(rgb - want).max(axis=2)
print("id mismatches", int((ids != leaf).sum()), "of", ids.size)
for thr in (1e-9, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1):
    print("diff > %g: %d" % (thr, int((d > thr).sum())))
bad = np.argwhere(d > 1e-4)
for x, y in bad[:25]:
    print(x, y, "gpu", rgb[x, y], "want", want[x, y], "id", ids[x, y], leaf[x, y])
kinds = {}
for x, y in bad:
    kinds[int(leaf[x, y])] = kinds.get(int(leaf[x, y]), 0) + 1
print("by leaf", kinds)
