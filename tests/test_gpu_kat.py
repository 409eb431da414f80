"""The reference's own edge-ray tables through the HIP path.

tests/test_oracle_kat.py holds the ORACLE to the known-answer rows of the reference's doctests; here the same rays --
cone shapes.py:996-1039, cylinder :1047-1141, cube :1150-1193, plane :1212-1234, sphere :1301-1358 -- plus the edge rays
random worlds never produce go through librtc_hip.so: 1x1 cameras whose inverse transform states the ray exactly
(origin = its translation, direction = minus its third column, rtc_camera.inv_transform), once with the object alone
(single-leaf route) and once among twelve others (BVH route), hit / miss, hit id and colour against the oracle in fp64.
These are the axis-aligned, on-the-slab and a ~ 0 rays that the fp32 box tests in front of the leaf tests must not cull."""
import numpy as np
import pytest

from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd import shapes as S
from ray_tracer_challenge_amd.canvas import color
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.matrix import rotation_z, scaling, translation
from ray_tracer_challenge_amd.renderer import World, point_light
from ray_tracer_challenge_amd.vector import point

pytestmark = pytest.mark.gpu


class RayCamera(object):
    """1 x 1 pixels; ray_for_pixel(0, 0) (renderer.py:742-751) = (origin, normalize(direction)) exactly: world_x = world_y = 0,
    pixel = inv * (0, 0, -1) = origin + direction."""
    hsize = vsize = 1
    half_width = half_height = 1.0
    pixel_size = 2.0
    field_of_view = 0.0

    def __init__(self, origin, direction):
        o, d = np.asarray(origin, float), np.asarray(direction, float)
        self.inv_transform = np.array([[1, 0, -d[0], o[0]], [0, 1, -d[1], o[1]], [0, 0, -d[2], o[2]], [0, 0, 0, 1.0]])


def _mat(r, g, b):
    return S.Material(color=color(r, g, b), ambient=0.2, diffuse=0.7, specular=0.3, shininess=50.0)


def _shapes():
    I = None
    unit = {
        "sphere": lambda: S.Sphere(_mat(1, 0.2, 0.2), I),
        "sphere2": lambda: S.Sphere(_mat(1, 0.2, 0.2), scaling(2, 2, 2)),
        "sphere5": lambda: S.Sphere(_mat(1, 0.2, 0.2), translation(5, 0, 0)),
        "plane": lambda: S.Plane(_mat(0.2, 1, 0.2), I),
        "cube": lambda: S.Cube(_mat(0.2, 0.2, 1), I),
        "cylinder": lambda: S.Cylinder(_mat(1, 1, 0.2), I),
        "cylinder12": lambda: S.Cylinder(_mat(1, 1, 0.2), I, minimum=1.0, maximum=2.0, closed=False),
        "cylinder12c": lambda: S.Cylinder(_mat(1, 1, 0.2), I, minimum=1.0, maximum=2.0, closed=True),
        "cone": lambda: S.Cone(_mat(0.2, 1, 1), I),
        "conec": lambda: S.Cone(_mat(0.2, 1, 1), I, minimum=-0.5, maximum=0.5, closed=True),
        "cone02": lambda: S.Cone(_mat(0.2, 1, 1), I, minimum=0.0, maximum=2.0, closed=False),
        "triangle": lambda: S.Triangle(_mat(1, 0.2, 1), I, point(0, 1, 0), point(-1, 0, 0), point(1, 0, 0)),
    }
    return unit


def n(*v):
    v = np.asarray(v, float)
    return v / np.sqrt((v * v).sum())


s5 = np.sqrt(0.5)
# (shape, origin, direction): rows of the reference's doctest tables (test_oracle_kat.py has their expected t values) ...
ROWS = [
    ("sphere", (0, 0, -5), (0, 0, 1)), ("sphere", (0, 1, -5), (0, 0, 1)), ("sphere", (0, 2, -5), (0, 0, 1)), ("sphere", (0, 0, 0), (0, 0, 1)),
    ("sphere", (0, 0, 5), (0, 0, 1)), ("sphere2", (0, 0, -5), (0, 0, 1)), ("sphere5", (0, 0, -5), (0, 0, 1)),
    ("plane", (0, 10, 0), (0, 0, 1)), ("plane", (0, 0, 0), (0, 0, 1)), ("plane", (0, 1, 0), (0, -1, 0)), ("plane", (0, -1, 0), (0, 1, 0)),
    ("cube", (5, 0.5, 0), (-1, 0, 0)), ("cube", (-5, 0.5, 0), (1, 0, 0)), ("cube", (0.5, 5, 0), (0, -1, 0)), ("cube", (0.5, -5, 0), (0, 1, 0)),
    ("cube", (0.5, 0, 5), (0, 0, -1)), ("cube", (0.5, 0, -5), (0, 0, 1)), ("cube", (0, 0.5, 0), (0, 0, 1)),
    ("cube", (-2, 0, 0), (0.2673, 0.5345, 0.8018)), ("cube", (0, -2, 0), (0.8018, 0.2673, 0.5345)), ("cube", (0, 0, -2), (0.5345, 0.8018, 0.2673)),
    ("cube", (2, 0, 2), (0, 0, -1)), ("cube", (0, 2, 2), (0, -1, 0)), ("cube", (2, 2, 0), (-1, 0, 0)),
    ("cylinder", (1, 0, 0), (0, 1, 0)), ("cylinder", (0, 0, 0), (0, 1, 0)), ("cylinder", (0, 0, -5), n(1, 1, 1)),
    ("cylinder", (1, 0, -5), (0, 0, 1)), ("cylinder", (0, 0, -5), (0, 0, 1)), ("cylinder", (0.5, 0, -5), n(0.1, 1, 1)),
    ("cylinder12", (0, 1.5, 0), n(0.1, 1, 0)), ("cylinder12", (0, 3, -5), (0, 0, 1)), ("cylinder12", (0, 0, -5), (0, 0, 1)),
    ("cylinder12", (0, 2, -5), (0, 0, 1)), ("cylinder12", (0, 1, -5), (0, 0, 1)), ("cylinder12", (0, 1.5, -2), (0, 0, 1)),
    ("cylinder12c", (0, 3, 0), (0, -1, 0)), ("cylinder12c", (0, 3, -2), n(0, -1, 2)), ("cylinder12c", (0, 4, -2), n(0, -1, 1)),
    ("cylinder12c", (0, 0, -2), n(0, 1, 2)), ("cylinder12c", (0, -1, -2), n(0, 1, 1)),
    ("cone", (0, 0, -5), (0, 0, 1)), ("cone", (0, 0, -5), n(1, 1, 1)), ("cone", (1, 1, -5), n(-0.5, -1, 1)), ("cone", (0, 0, -1), n(0, 1, 1)),
    ("conec", (0, 0, -5), (0, 1, 0)), ("conec", (0, 0, -0.25), n(0, 1, 1)), ("conec", (0, 0, -0.25), (0, 1, 0)),
]
# ... and by hand, the degenerate rays of each leaf test
ROWS += [
    # cube: a zero direction component with the origin ON that slab's plane -- (lo - o) * inf = 0 * inf = NaN, which the
    # reference's max()/min() chains let through or not by position (shapes.py:606-614); edges, a corner, inside on a face
    ("cube", (1, 0.5, -5), (0, 0, 1)), ("cube", (-1, 0.5, -5), (0, 0, 1)), ("cube", (0.5, 1, -5), (0, 0, 1)), ("cube", (0.5, -1, -5), (0, 0, 1)),
    ("cube", (1, 1, -5), (0, 0, 1)), ("cube", (-1, -1, -5), (0, 0, 1)), ("cube", (1, 0, 0), (0, 1, 0)), ("cube", (0, 0, 1), (1, 0, 0)),
    ("cube", (-5, 1, 1), (1, 0, 0)), ("cube", (1, -5, 0.25), (0, 1, 0)), ("cube", (0, 0, 0), (0, 0, 1)), ("cube", (0, 0, -5), (0, 1e-7, 1)),
    ("cube", (5, 5, 5), n(-1, -1, -1)), ("cube", (1 + 1e-9, 0, -5), (0, 0, 1)), ("cube", (1 - 1e-9, 0, -5), (0, 0, 1)),
    # cone: a ~ 0 with b != 0 (one hit, even outside a truncated cone's bounds, shapes.py:647-650), a ~ 0 and b ~ 0 (0 / 0)
    ("cone02", (0, 0, -1), n(0, 1, 1)), ("cone02", (0, -3, -1), n(0, 1, 1)), ("cone", (0, 0, 0), n(0, 1, 1)), ("cone", (1, 1, 0), n(1, 1, 0)),
    ("cone", (0, 0, -2), n(0, -1, 1)), ("conec", (0, 0, 0), n(1, 1, 0)), ("cone", (0, 0, 0), (0, 1, 0)), ("cone", (0, 5, 0), (0, -1, 0)),
    # cone caps accept x^2 + z^2 <= |y| (shapes.py:689-693): the brim between the cone's radius 0.5 and sqrt(0.5), its rim, outside
    ("conec", (0.6, 2, 0), (0, -1, 0)), ("conec", (s5, 2, 0), (0, -1, 0)), ("conec", (0.5, 2, 0.5), (0, -1, 0)), ("conec", (0.71, 2, 0), (0, -1, 0)),
    ("conec", (0.6, -2, 0), (0, 1, 0)), ("conec", (0.2, 0, -3), (0, 0, 1)), ("conec", (0, 0.5, -3), (0, 0, 1)), ("conec", (0, -0.5, -3), (0, 0, 1)),
    # cylinder: parallel to the axis on the wall (a ~ 0), through the cap rim, along a cap plane, tangent to the wall
    ("cylinder12c", (1, 0, 0), (0, 1, 0)), ("cylinder12c", (0.999999999, 0, 0), (0, 1, 0)), ("cylinder12c", (0, 2, -5), (0, 0, 1)),
    ("cylinder12c", (0, 1, -5), (0, 0, 1)), ("cylinder12c", (1, 1.5, -5), (0, 0, 1)), ("cylinder12c", (0, 1.5, 0), (0, 1, 0)),
    ("cylinder12c", (0, 1.5, 0), (1, 0, 0)), ("cylinder", (1, 0, -5), n(0, 1, 1)),
    # sphere: tangent, origin on the surface, grazing just inside / outside
    ("sphere", (1, 0, -5), (0, 0, 1)), ("sphere", (0, 0, -1), (0, 0, 1)), ("sphere", (0, 0, 1), (0, 0, 1)), ("sphere", (1 - 1e-12, 0, -5), (0, 0, 1)),
    ("sphere", (1 + 1e-12, 0, -5), (0, 0, 1)),
    # plane: parallel above, in and just off the plane; |d.y| at the reference's EPSILON
    ("plane", (0, 1, 0), (1, 0, 0)), ("plane", (0, 1, 0), n(1, -1e-6, 0)), ("plane", (0, 1, 0), n(1, -0.999e-6, 0)), ("plane", (0, 1, 0), n(1, -1.001e-6, 0)),
    # triangle: through each vertex and edge (u = 0, v = 0, u + v = 1), parallel to its plane, from behind
    ("triangle", (0, 1, -2), (0, 0, 1)), ("triangle", (-1, 0, -2), (0, 0, 1)), ("triangle", (1, 0, -2), (0, 0, 1)), ("triangle", (0, 0, -2), (0, 0, 1)),
    ("triangle", (-0.5, 0.5, -2), (0, 0, 1)), ("triangle", (0.5, 0.5, -2), (0, 0, 1)), ("triangle", (0, 0.5, -2), (0, 1, 0)), ("triangle", (0, 0.5, 2), (0, 0, -1)),
    ("triangle", (0, 0.5, -2), (0, 0, 1)), ("triangle", (0, 1 + 1e-12, -2), (0, 0, 1)), ("triangle", (0, 1 - 1e-12, -2), (0, 0, 1)),
]


def _others():
    """twelve more objects well away from the unit shapes (the rays above may still meet some: the oracle says which)"""
    out = []
    for k in range(12):
        a = 2 * np.pi * k / 12
        xf = translation(9 * np.cos(a), 3 * np.sin(3 * a), 9 * np.sin(a)) * rotation_z(0.3 * k) * scaling(0.7, 0.9, 0.8)
        cls = [S.Sphere, S.Cube, S.Sphere][k % 3]
        out.append(cls(_mat(0.5, 0.5, 0.1 + 0.07 * k), xf))
    return out


@pytest.mark.parametrize("crowd", [False, True], ids=["alone", "among12"])
def test_reference_edge_rays_through_the_kernel(crowd, oracle_cls):
    lights = [point_light(point(-10, 10, -10), color(1, 1, 1)), point_light(point(6, -8, 7), color(0.4, 0.4, 0.4))]
    shapes = _shapes()
    checked, hits = 0, 0
    disagreements = []
    for name in sorted(set(r[0] for r in ROWS)):
        obj = shapes[name]()
        world = World(lights, [obj] + (_others() if crowd else []))
        orc = oracle_cls(world)
        with _ffi.Scene(compile_scene(world), device=0) as sc:
            for shape, o, d in ROWS:
                if shape != name:
                    continue
                cam = RayCamera(o, d)
                want, leaf, t, _ = orc.render_pixels(cam, [0], [0])
                got, ids = sc.render(cam, precision="fp64", want_ids=True)
                checked += 1
                hits += int(leaf[0] >= 0)
                same = (ids[0, 0] == leaf[0]) and (np.allclose(got[0, 0], want[0], rtol=0, atol=1e-9) or
                                                     (not np.isfinite(want[0]).all() and not np.isfinite(got[0, 0]).all()))
                if not same:
                    disagreements.append((name, o, tuple(np.round(d, 6)), int(ids[0, 0]), int(leaf[0]), got[0, 0].tolist(), want[0].tolist()))
    print("%d rays (%d hit something), %d disagreements" % (checked, hits, len(disagreements)))
    assert checked == len(ROWS) and hits > checked // 3
    # Accepted deviation, named: a cone ray exactly parallel to its slope (|a| <= 1e-8) gets the reference's single hit
    # t = -c / 2b even outside a truncated cone's bounds (shapes.py:647-650); the kernel's bounding boxes are built from the
    # cone's bounds and may cull that hit.  Row ("cone02", (0, -3, -1), n(0, 1, 1)) is that ray; nothing else may differ.
    allowed = [x for x in disagreements if x[0] == "cone02" and x[1] == (0, -3, -1)]
    assert [x for x in disagreements if x not in allowed] == [], disagreements
