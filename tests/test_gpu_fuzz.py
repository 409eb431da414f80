"""Random scenes: the HIP path against the CPU oracle on seeded worlds the fixtures do not cover -- every primitive kind
under random rotate/scale/shear/translate chains, groups, CSG, reflective and refractive materials with mixed refractive
indices (the n1/n2 container pass), shadow-less objects, every pattern family including perturbed ones, point and area
lights -- at object counts that take the single-object, flat-list (2-4 objects) and BVH routes through the kernel.

The oracle is pinned to the reference by tests/test_oracle_golden.py and tests/test_oracle_kat.py; here it is the checker
for inputs the reference never rendered.  Discrete decisions (silhouettes, pattern boundaries, equal-t ties) may fall
differently within rounding, so a small fraction of pixels may differ; everywhere else fp64 agrees to 1e-9.
RTC_FUZZ_SEEDS=400 runs more worlds than the default 48 (measured: 395 of 400 with no fp64 pixel off at all, the rest <= 5
of 3 072).  fp32 has its own test over 200 worlds, test_fp32_gate_over_random_worlds: north_star's gate (at most 0.1 % of
the pixels more than one 8-bit level off) holds on 97.8 % of 404 worlds measured, and the worlds beyond it are named there
with their causes.  Two defects this test found and that are fixed: the n1/n2 container pass counted the hit as its own predecessor when its t was recomputed one ulp lower (glass
objects lost their refraction in fp32), and fp32 cap points of cylinders/cones fell outside the reference's 1e-6 cap test.
"""
import os

import numpy as np
import pytest

from conftest import explain_mismatches, frac_off_by_more_than_one, move_triangles, record_classes
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd import shapes as S
from ray_tracer_challenge_amd.canvas import color
from ray_tracer_challenge_amd.compile import compile_scene
from ray_tracer_challenge_amd.matrix import rotation_x, rotation_y, rotation_z, scaling, shearing, translation
from ray_tracer_challenge_amd.renderer import AreaLight, Camera, World, point_light, view_transform
from ray_tracer_challenge_amd.vector import point, vector

pytestmark = pytest.mark.gpu

HSIZE, VSIZE = 64, 48


def bind(mods=None):
    """Which constructors the world generators below use: this package's (default), or -- `mods` = the dict
    oracle/ref_harness.import_reference() returns -- the reference's own classes and functions, which have the same names and
    signatures.  oracle/make_golden.py builds the SAME random worlds (same seeds, same draws) out of the reference's classes
    that way and renders them with the reference: tests/golden/fuzz_worlds.npz."""
    g = globals()
    if mods is None:
        import ray_tracer_challenge_amd.canvas as c, ray_tracer_challenge_amd.matrix as m, ray_tracer_challenge_amd.renderer as r
        import ray_tracer_challenge_amd.shapes as sh, ray_tracer_challenge_amd.vector as v
    else:
        c, m, r, sh, v = mods["canvas"], mods["matrix"], mods["renderer"], mods["shapes"], mods["vector"]
    g["S"], g["color"], g["point"], g["vector"] = sh, c.color, v.point, v.vector
    for name in ("rotation_x", "rotation_y", "rotation_z", "scaling", "shearing", "translation"):
        g[name] = getattr(m, name)
    for name in ("AreaLight", "Camera", "World", "point_light", "view_transform"):
        g[name] = getattr(r, name)
    if mods is None:
        g["_texture_canvas"] = lambda pixels: c.TextureCanvas(pixels=pixels)
    else:                                  # the reference's TextureCanvas reads a file: the same texels, through a PNG
        import tempfile
        from PIL import Image
        tmp = tempfile.mkdtemp(prefix="rtc_fuzz_tex_")

        def from_file(pixels):
            path = os.path.join(tmp, "t%d.png" % len(os.listdir(tmp)))
            Image.fromarray(pixels).save(path)
            return c.TextureCanvas(path)
        g["_texture_canvas"] = from_file

_SPREAD = [2.5]          # how far from the origin objects are scattered (smaller for worlds of few objects, so that they are in view)


bind(None)


def _identity():
    return translation(0, 0, 0)          # (the reference's shapes have no default for a missing transform)


def _rgb(rng, lo=0.05, hi=1.0):
    return color(*rng.uniform(lo, hi, 3))


def _transform(rng):
    m = translation(*rng.uniform(-_SPREAD[0], _SPREAD[0], 3))
    m = m * rotation_y(rng.uniform(0, 6.28)) * rotation_x(rng.uniform(0, 6.28)) * rotation_z(rng.uniform(0, 6.28))
    if rng.random() < 0.25:
        m = m * shearing(*rng.uniform(-0.4, 0.4, 6))
    return m * scaling(*rng.uniform(0.35, 1.3, 3))


def _small_transform(rng):
    return translation(*rng.uniform(-0.5, 0.5, 3)) * rotation_y(rng.uniform(0, 6.28)) * scaling(*rng.uniform(0.2, 1.5, 3))


def _simple_pattern(rng):
    cls = [S.Checkers, S.Gradient, S.RadialGradient, S.Ring, S.Stripe][rng.integers(5)]
    return cls(_rgb(rng), _rgb(rng), _small_transform(rng))


def _uv_pattern(rng, images=False):
    r = rng.random()
    if images and r < 0.5:                               # an image texture (earth, skybox): random texels, odd sizes
        h, w = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        return S.UVTexturePattern(_texture_canvas(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)))
    if r < 0.5:
        return S.UVChecker(_rgb(rng), _rgb(rng), int(rng.integers(2, 9)), int(rng.integers(2, 9)))
    return S.UVAlignCheck(_rgb(rng), _rgb(rng), _rgb(rng), _rgb(rng), _rgb(rng))


def _pattern(rng, depth=0):
    r = rng.random()
    if depth >= 2 or r < 0.35:
        return _simple_pattern(rng)
    if r < 0.47:
        return S.BlendedPattern(_pattern(rng, depth + 1), _pattern(rng, depth + 1))
    if r < 0.59:
        return S.NestedPattern(_simple_pattern(rng), _pattern(rng, depth + 1), _pattern(rng, depth + 1))
    if r < 0.74:
        return S.PerturbedPattern(_pattern(rng, depth + 2), frequency=float(rng.uniform(0.15, 0.8)),
                                  scale_factor=float(rng.uniform(0.05, 0.4)), octaves=int(rng.integers(1, 5)))
    if r < 0.84:
        return S.TextureMapPattern(_uv_pattern(rng), [S.PlaneUVMap, S.SphereUVMap][rng.integers(2)], _small_transform(rng))
    if r < 0.92:
        return S.CylinderMapPattern(_uv_pattern(rng), _uv_pattern(rng), S.CylinderUVMap, _small_transform(rng))
    return S.CubeMapPattern(*[_uv_pattern(rng) for _ in range(6)], S.CubeUVMap, _small_transform(rng))


def _material(rng):
    m = S.Material(color=_rgb(rng), ambient=rng.uniform(0.05, 0.3), diffuse=rng.uniform(0.4, 0.9), specular=rng.uniform(0, 0.9),
                   shininess=float(rng.choice([10.0, 50.0, 200.0, 300.0])))
    if rng.random() < 0.3:
        m.reflective = np.float64(rng.uniform(0.2, 0.9))
    if rng.random() < 0.3:
        m.transparency = np.float64(rng.uniform(0.3, 0.95))
        m.refractive_index = np.float64(rng.choice([1.0, 1.33, 1.5, 2.417]))
    if rng.random() < 0.15:
        m.casts_shadow = False
    if rng.random() < 0.45:
        m.pattern = _pattern(rng)
    return m


def _primitive(rng, allow_unbounded=True):
    k = rng.integers(7)
    mat, xf = _material(rng), _transform(rng)
    if k == 0:
        return S.Sphere(mat, xf)
    if k == 1:
        return S.Cube(mat, xf)
    if k in (2, 3):
        cls = S.Cylinder if k == 2 else S.Cone
        if allow_unbounded and rng.random() < 0.1:
            return cls(mat, xf)                                            # infinite: goes to the unbounded list
        lo = rng.uniform(-1.5, 0.2)
        return cls(mat, xf, minimum=lo, maximum=lo + rng.uniform(0.4, 1.8), closed=bool(rng.random() < 0.6))
    p = [point(*rng.uniform(-1.2, 1.2, 3)) for _ in range(3)]
    if k == 4:
        return S.Triangle(mat, xf, *p)
    if k == 5:
        n = [vector(*(v / np.linalg.norm(v))) for v in rng.normal(size=(3, 3))]
        return S.SmoothTriangle(mat, xf, *p, *n)
    return S.Sphere(mat, xf)


def _object(rng, budget):
    r = rng.random()
    if r < 0.15 and budget["csg"] > 0:
        budget["csg"] -= 1
        op = [S.CSGUnion, S.CSGIntersect, S.CSGDifference][rng.integers(3)]
        left, right = _primitive(rng, False), _primitive(rng, False)
        right.transform = left.transform * translation(*rng.uniform(-0.6, 0.6, 3)) * scaling(*rng.uniform(0.6, 1.2, 3))   # overlapping
        return S.CSG(_material(rng), _small_transform(rng), op, left, right)
    if r < 0.3:
        return S.Group(_material(rng), _small_transform(rng), [_primitive(rng) for _ in range(int(rng.integers(2, 4)))])
    return _primitive(rng)


def _exact(what, orc, cam, rgb, want, ids=None, leaf=None, ok=None, remaining=5, allowed=0.004, tol=1e-9):
    """fp64 against the oracle: every differing pixel must be a tie, a grazing hit, acne or on an edge (SURVEY.md 8d,
    conftest.explain_mismatches) -- none unexplained -- and those are few."""
    w, h = rgb.shape[0], rgb.shape[1]
    xs, ys = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    sel = np.ones((w, h), bool) if ok is None else ok
    unexplained, classes = explain_mismatches(orc, cam, xs[sel], ys[sel], rgb[sel], want[sel], None if ids is None else ids[sel],
                                              None if leaf is None else leaf[sel], remaining=remaining, tol=tol)
    print("%s: differing and explained %s, unexplained %d" % (what, classes, len(unexplained)))
    if "tree" in what:
        record_classes("fuzz: " + what, int(sel.sum()), classes, len(unexplained))
    assert not unexplained, "%s: pixels that differ from the oracle without being tie / grazing / acne / edge: %s" % (what, unexplained[:12])
    assert sum(classes.values()) <= allowed * sel.sum() + 1


def random_world(seed):
    rng = np.random.default_rng(seed)
    n_objects = [1, 2, 3, 4, 7, 12][seed % 6]
    budget = {"csg": 2}
    _SPREAD[0] = 0.8 if n_objects <= 2 else (1.6 if n_objects <= 4 else 2.5)
    objs = [_object(rng, budget) for _ in range(n_objects)]
    if rng.random() < 0.6:                                                 # a floor
        floor = S.Plane(_material(rng), translation(0, -2.6, 0))
        floor.material.transparency = np.float64(0.0)
        objs.append(floor)
    lights = []
    if seed % 3 == 0:
        lights.append(AreaLight(point(*rng.uniform(-4, 4, 3)) + vector(0, 7, 0), vector(1.5, 0, 0), 3, vector(0, 0.5, 1.5), 3, False, _rgb(rng, 0.5, 1.0)))
    for _ in range(int(rng.integers(1, 3)) if seed % 3 else 1 if rng.random() < 0.5 else 0):
        lights.append(point_light(point(*(rng.uniform(-6, 6, 3) + np.array([0, 6, -4]))), _rgb(rng, 0.3, 1.0)))
    eye = point(*(rng.uniform(-1, 1, 3) + np.array([0, 1.5, -8.5 if n_objects > 4 else -5.0])))
    cam = Camera(HSIZE, VSIZE, 0.95, view_transform(eye, point(0, -0.3, 0), vector(0, 1, 0)))
    return cam, World(lights, objs)


# the worlds that exposed the defects named above stay in the default run: 110 / 215 (container pass, fp32 glass), 129 (fp32 cap
# normals), 2004 / 1595 / 2267 / 2393 (the cone's cap brim outside the reference's own bounds)
REGRESSIONS = [110, 129, 215, 1595, 2004, 2267, 2393]


@pytest.mark.parametrize("seed", sorted(set(range(int(__import__("os").environ.get("RTC_FUZZ_SEEDS", "48")))) | set(REGRESSIONS)))
def test_random_world_matches_oracle(seed, oracle_cls):
    cam, world = random_world(seed)
    xs, ys = np.meshgrid(np.arange(HSIZE), np.arange(VSIZE), indexing="ij")
    px, py = xs.ravel(), ys.ravel()
    orc = oracle_cls(world)
    want, leaf, t_hit, rays = orc.render_pixels(cam, px, py)
    want, leaf = want.reshape(HSIZE, VSIZE, 3), leaf.reshape(HSIZE, VSIZE)
    near = (~(np.isfinite(t_hit) & (t_hit > 60.0))).reshape(HSIZE, VSIZE)      # fp32 is not asked to resolve patterns on a floor hundreds of units away
    with _ffi.Scene(compile_scene(world), device=0) as sc:
        st = {}
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
        f32, ids32 = sc.render(cam, precision="fp32", want_ids=True)
    assert np.isfinite(want).all() == np.isfinite(rgb).all()
    ok = np.isfinite(want).all(axis=2) & np.isfinite(rgb).all(axis=2)
    off = (np.abs(rgb - want).max(axis=2) > 1e-9) & ok
    print("seed %d: fp64 off %d, ids %d, fp32 off>1 %.4f, ids32 %d of %d px" % (seed, off.sum(), (ids != leaf).sum(),
          frac_off_by_more_than_one(f32[ok & near], want[ok & near]), (ids32 != leaf).sum(), off.size))      # shown with -s (tools/fuzz_diag.py digs further)
    _exact("seed %d" % seed, orc, cam, rgb, want, ids, leaf, ok)
    # rays kind by kind: equal up to the few pixels whose discrete decisions fell differently
    got = [st["rays"][0], st["rays"][1] + st["shadow_rays_culled"], st["rays"][2], st["rays"][3]]
    for g, w in zip(got, [int(v) for v in rays]):
        assert abs(g - w) <= 0.02 * max(w, 1) + 24
    assert frac_off_by_more_than_one(f32[ok & near], want[ok & near]) <= FP32_WORST   # world by world; the gate itself: test_fp32_gate_over_random_worlds
    assert (ids32 != leaf).mean() <= 0.02


# fp32 against north_star's gate.  A 64 x 48 frame has 3 072 pixels, so "at most 0.1 % of the pixels more than one 8-bit
# level off" means at most 3 pixels.  Measured over seeds 0..399 + REGRESSIONS (profiles/r02_fp32_fuzz.txt): 97.8 % of the
# worlds are within the gate, 99.3 % within twice the gate, the worst is seed 64.  Which worlds sit at 4 or 5 pixels moves
# with the last bits of the build (fused multiply-adds); the ones seen beyond the gate among the first 200 seeds, with what
# tools/fuzz_diag.py (EPS_SWEEP=1) found on them:
FP32_BEYOND_GATE = {
    1:   "a lone SmoothTriangle whose random vertex normals lie almost in its plane: the 1e-5 offset along such a normal lifts the shadow "
         "ray's origin less than an fp32 ulp of the hit point off the plane, so 5-7 of its 74 px self-shadow differently (a 1e-4 offset: 0 px)",
    55:  "the same: a lone mirror SmoothTriangle with near-tangent vertex normals, 5-7 of 67 px",
    64:  "a diamond-index (2.417) glass cylinder over a mirror floor: all but one of the 10-15 px appear at depth >= 2, on refracted rays "
         "that leave the cylinder at grazing angles where total internal reflection flips within fp32 rounding (depth 0: none off)",
    92:  "4-5 px on a glass sphere with a perturbed pattern at depth >= 2; 2 of them move in fp64 too when the offset goes from 1e-6 to 1e-5",
    155: "4-5 px inside a large transparent cone, depth >= 1; 2 of them move in fp64 too when the offset goes from 1e-6 to 1e-5 "
         "(was 1.7 % of the frame before the cone's conditioned solve)",
    161: "5 px at the terminator of a sphere under two point lights (shadow offset 1e-5 against the reference's 1e-6)",
}
FP32_GATE, FP32_WORST = 0.001, 0.006


def test_fp32_gate_over_random_worlds(oracle_cls):
    """fp32 frames of 200 random worlds against the fp64 oracle: at least 95 % of the worlds within north_star's gate, none
    beyond 0.6 %, and every world beyond TWICE the gate is one of those named above (a new one is a regression to look at;
    between one and two times the gate a world is one pixel away from passing, and which ones are moves with the build)."""
    xs, ys = np.meshgrid(np.arange(HSIZE), np.arange(VSIZE), indexing="ij")
    share = {}
    for seed in range(200):
        cam, world = random_world(seed)
        want, _, t_hit, _ = oracle_cls(world).render_pixels(cam, xs.ravel(), ys.ravel())
        want = want.reshape(HSIZE, VSIZE, 3)
        near = (~(np.isfinite(t_hit) & (t_hit > 60.0))).reshape(HSIZE, VSIZE) & np.isfinite(want).all(axis=2)
        with _ffi.Scene(compile_scene(world), device=0) as sc:
            f32, _ = sc.render(cam, precision="fp32")
        share[seed] = frac_off_by_more_than_one(f32[near], want[near])
    v = np.array(list(share.values()))
    beyond = sorted(k for k, f in share.items() if f > FP32_GATE + 1e-4)            # 3 of 3 072 px prints as 0.00098
    print("fp32 over %d worlds: %.1f %% within the gate, worst %.4f (seed %d); beyond the gate: %s" % (
        len(v), 100 * (v <= FP32_GATE + 1e-4).mean(), v.max(), int(np.argmax(v)), {k: round(share[k], 4) for k in beyond}))
    assert (v <= FP32_GATE + 1e-4).mean() >= 0.95
    assert v.max() <= FP32_WORST
    far_beyond = {k for k, f in share.items() if f > 2 * FP32_GATE + 1e-4}
    assert far_beyond <= set(FP32_BEYOND_GATE), "worlds newly beyond twice the fp32 gate: %s" % sorted(far_beyond - set(FP32_BEYOND_GATE))


def _fuzz_golden():
    import test_oracle_fuzz_golden as G
    return G


@pytest.mark.parametrize("kind,i", _fuzz_golden().WORLDS)
def test_random_world_matches_the_reference(kind, i, oracle_cls):
    """The kernel against the REFERENCE's own frames of these worlds (tests/golden/fuzz_worlds.npz: the same worlds built out of
    the reference's classes and rendered by it -- general, nested-CSG and mesh worlds; tests/test_oracle_fuzz_golden.py holds
    the oracle to them as well)."""
    G = _fuzz_golden()
    seed, cam, world = G.build(kind, i)
    want, ref_t = G.RGB[kind][i], G.HIT_T[kind][i]
    near = ~(np.isfinite(ref_t) & (ref_t > 60.0))          # fp32 is not asked to resolve patterns on a floor hundreds of units away
    with _ffi.Scene(compile_scene(world), device=0) as sc:
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
        f32, _ = sc.render(cam, precision="fp32")
    assert np.array_equal(np.isfinite(want).all(axis=2), np.isfinite(rgb).all(axis=2))
    ok = np.isfinite(want).all(axis=2)
    _exact("%s seed %d against the reference" % (kind, seed), oracle_cls(world), cam, rgb, want, None, None, ok)
    assert frac_off_by_more_than_one(f32[ok & near], want[ok & near]) <= FP32_WORST


def _mesh(rng, n_u, n_v, smooth):
    """A bumpy tessellated ellipsoid as a group of (smooth) triangles under its own transform."""
    radii = rng.uniform(0.6, 1.4, 3)
    bump = rng.uniform(0.0, 0.15)

    def vertex(i, j):
        th, ph = np.pi * i / n_u, 2 * np.pi * (j % n_v) / n_v
        r = 1.0 + bump * np.sin(3 * th + 1.3) * np.cos(2 * ph)
        return radii * r * np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])

    def normal(p):
        n = p / (radii * radii)
        return vector(*(n / np.linalg.norm(n)))
    mat = _material(rng)
    tris = []
    for i in range(n_u):
        for j in range(n_v):
            quad = [vertex(i, j), vertex(i + 1, j), vertex(i + 1, j + 1), vertex(i, j + 1)]
            for a, b, c in ((0, 1, 2), (0, 2, 3)):
                pa, pb, pc = quad[a], quad[b], quad[c]
                if np.linalg.norm(np.cross(pb - pa, pc - pa)) < 1e-9:
                    continue                                             # the degenerate triangles at the poles
                if smooth:
                    tris.append(S.SmoothTriangle(mat, _identity(), point(*pa), point(*pb), point(*pc), normal(pa), normal(pb), normal(pc)))
                else:
                    tris.append(S.Triangle(mat, _identity(), point(*pa), point(*pb), point(*pc)))
    return S.Group(S.Material(), _transform(rng), tris)


def random_mesh_world(seed):
    rng = np.random.default_rng(1000 + seed)
    _SPREAD[0] = 1.2
    objs = [_mesh(rng, int(rng.integers(6, 12)), int(rng.integers(8, 16)), smooth=bool(seed % 2))]
    objs += [_primitive(rng) for _ in range(int(rng.integers(0, 4)))]
    floor = S.Plane(_material(rng), translation(0, -2.6, 0))
    floor.material.transparency = np.float64(0.0)
    objs.append(floor)
    lights = [point_light(point(*(rng.uniform(-5, 5, 3) + np.array([0, 7, -5]))), _rgb(rng, 0.5, 1.0))]
    if seed % 4 == 0:
        lights.append(AreaLight(point(-3, 6, -2), vector(1, 0, 0), 2, vector(0, 0, 1), 2, False, _rgb(rng, 0.3, 0.8)))
    w, h = 71, 45                                                          # not multiples of the 32-pixel tile or the 8-pixel block
    cam = Camera(w, h, 0.9, view_transform(point(0.5, 1.8, -6.0), point(0, 0, 0), vector(0, 1, 0)))
    return cam, World(lights, objs), rng


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RTC_FUZZ_MESH_SEEDS", "8"))))
def test_random_mesh_world(seed, oracle_cls):
    """Triangle-BVH route: a tessellated mesh (flat or smooth, possibly glass or mirror) among a few primitives -- host-built
    tree against the oracle, then the GPU-built tree and a refit of it against the host-built one, then a random subset of
    tiles against the same pixels of the full frame."""
    cam, world, rng = random_mesh_world(seed)
    w, h = cam.hsize, cam.vsize
    xs, ys = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    orc = oracle_cls(world)
    want, leaf, t_hit, _ = orc.render_pixels(cam, xs.ravel(), ys.ravel())
    want, leaf = want.reshape(w, h, 3), leaf.reshape(w, h)
    near = (~(np.isfinite(t_hit) & (t_hit > 60.0))).reshape(w, h)
    cs = compile_scene(world)
    with _ffi.Scene(cs, device=0) as sc:
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
        _exact("mesh seed %d" % seed, orc, cam, rgb, want, ids, leaf)
        f32, _ = sc.render(cam, precision="fp32")
        assert frac_off_by_more_than_one(f32[near], want[near]) <= 0.005
        tiles = np.sort(rng.choice(6, size=3, replace=False)).astype(np.int32)      # 3 x 2 tiles of 32 x 32 cover 71 x 45
        part, part_ids = sc.render_tiles(cam, tiles, precision="fp64", want_ids=True)
        canvas_rgb, canvas_ids = np.full((w, h, 3), -7.0), np.full((w, h), -7, np.int32)
        _ffi.scatter_tiles(canvas_rgb, canvas_ids, tiles, part, part_ids)
        for t in tiles:
            x0, y0 = (int(t) % 3) * 32, (int(t) // 3) * 32
            np.testing.assert_array_equal(canvas_rgb[x0:x0 + 32, y0:y0 + 32], rgb[x0:x0 + 32, y0:y0 + 32])
            np.testing.assert_array_equal(canvas_ids[x0:x0 + 32, y0:y0 + 32], ids[x0:x0 + 32, y0:y0 + 32])
        assert (canvas_ids == -7).sum() == w * h - sum(min(32, w - (int(t) % 3) * 32) * min(32, h - (int(t) // 3) * 32) for t in tiles)
        sc.build_bvh_on_device()
        g, gi = sc.render(cam, precision="fp64", want_ids=True)
        _exact("mesh seed %d, GPU-built tree" % seed, orc, cam, g, want, gi, leaf)           # the gate: the oracle, as for the host-built tree
        g32, _ = sc.render(cam, precision="fp32")
        assert frac_off_by_more_than_one(g32[near], want[near]) <= 0.005
        assert (np.abs(g - rgb).max(axis=2) > 1e-9).mean() <= 0.004 and (gi != ids).mean() <= 0.004      # extra: host tree against GPU tree
        sc.update_triangles(cs.arrays["tri_geom"], cs.arrays["tri_norm"], refit=True)      # same rows: same boxes, same frame
        r, ri = sc.render(cam, precision="fp64", want_ids=True)
        np.testing.assert_array_equal(r, g)
        # the mesh moved (every vertex by its own amount, object space): rebuilt and refitted trees against the oracle of the moved world
        moved_cam, moved_world, _ = random_mesh_world(seed)
        shake = np.random.default_rng(77 + seed)
        move_triangles(moved_world, lambda p: p + 0.25 * np.sin(2.0 * p[:, [2, 0, 1]]) + 0.01 * shake.standard_normal(p.shape))
        mcs = compile_scene(moved_world)
        morc = oracle_cls(moved_world)
        mwant, mleaf, _, _ = morc.render_pixels(cam, xs.ravel(), ys.ravel())
        mwant, mleaf = mwant.reshape(w, h, 3), mleaf.reshape(w, h)
        for refit in (True, False):
            sc.update_triangles(cs.arrays["tri_geom"], cs.arrays["tri_norm"], refit=False)      # back to the start, fresh GPU tree
            sc.update_triangles(mcs.arrays["tri_geom"], mcs.arrays["tri_norm"], refit=refit)
            m, mi = sc.render(cam, precision="fp64", want_ids=True)
            _exact("mesh seed %d moved, GPU tree %s" % (seed, "refitted" if refit else "rebuilt"), morc, cam, m, mwant, mi, mleaf)


def _csg_tree(rng, depth, anchor):
    """Nested CSG: operands are primitives, groups of primitives or further CSG nodes, all overlapping around `anchor`."""
    def operand(d):
        r = rng.random()
        if d > 0 and r < 0.35:
            return _csg_tree(rng, d - 1, anchor)
        if r < 0.5:
            g = S.Group(_material(rng), anchor * _small_transform(rng), [])
            for _ in range(2):
                c = _primitive(rng, False)
                c.transform = translation(*rng.uniform(-0.4, 0.4, 3)) * scaling(*rng.uniform(0.5, 1.1, 3))
                g.add_child(c)
            return g
        p = _primitive(rng, False)
        p.transform = anchor * translation(*rng.uniform(-0.5, 0.5, 3)) * rotation_y(rng.uniform(0, 6.28)) * scaling(*rng.uniform(0.6, 1.3, 3))
        return p
    op = [S.CSGUnion, S.CSGIntersect, S.CSGDifference][rng.integers(3)]
    return S.CSG(_material(rng), _identity(), op, operand(depth), operand(depth))


def random_csg_world(seed):
    rng = np.random.default_rng(5000 + seed)
    _SPREAD[0] = 1.0
    objs = []
    for k in range(int(rng.integers(1, 4))):
        anchor = translation(*rng.uniform(-1.6, 1.6, 3)) * rotation_x(rng.uniform(0, 6.28)) * rotation_z(rng.uniform(0, 6.28))
        objs.append(_csg_tree(rng, 2, anchor))                              # any size: big ones are evaluated without a hit list
    if not objs or rng.random() < 0.4:
        objs.append(_primitive(rng))
    if rng.random() < 0.7:
        floor = S.Plane(_material(rng), translation(0, -2.8, 0))
        floor.material.transparency = np.float64(0.0)
        objs.append(floor)
    lights = [point_light(point(*(rng.uniform(-5, 5, 3) + np.array([0, 7, -5]))), _rgb(rng, 0.5, 1.0))]
    cam = Camera(HSIZE, VSIZE, 0.9, view_transform(point(0.3, 2.5, -6.5), point(0, -0.2, 0), vector(0, 1, 0)))
    return cam, World(lights, objs)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RTC_FUZZ_SEEDS", "16"))))
def test_random_csg_world(seed, oracle_cls):
    """CSG route: one to three nested CSG trees (operands: primitives, groups, CSG), transparent and reflective parts,
    a floor and sometimes an extra primitive, so that both CSG-only (single item) and mixed scenes occur."""
    cam, world = random_csg_world(seed)
    xs, ys = np.meshgrid(np.arange(HSIZE), np.arange(VSIZE), indexing="ij")
    orc = oracle_cls(world)
    want, leaf, t_hit, rays = orc.render_pixels(cam, xs.ravel(), ys.ravel())
    want, leaf = want.reshape(HSIZE, VSIZE, 3), leaf.reshape(HSIZE, VSIZE)
    near = (~(np.isfinite(t_hit) & (t_hit > 60.0))).reshape(HSIZE, VSIZE)
    with _ffi.Scene(compile_scene(world), device=0) as sc:
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
        f32, _ = sc.render(cam, precision="fp32")
    off = np.abs(rgb - want).max(axis=2) > 1e-9
    print("csg seed %d: fp64 off %d, ids %d, fp32 off>1 %.4f" % (seed, off.sum(), (ids != leaf).sum(), frac_off_by_more_than_one(f32[near], want[near])))
    _exact("csg seed %d" % seed, orc, cam, rgb, want, ids, leaf)
    assert frac_off_by_more_than_one(f32[near], want[near]) <= 0.004       # 400 CSG worlds: 98.5 % within 0.001, worst 0.0022


def _glass(rng):
    m = _material(rng)
    m.transparency, m.refractive_index, m.reflective = np.float64(0.7), np.float64(rng.choice([1.33, 1.5])), np.float64(0.2)
    m.pattern = None
    return m


LARGE_CSG_CASES = ["groups_of_spheres", "many_cylinders", "mesh_minus_sphere", "glass_clusters"]


def large_csg_world(case):
    """`case`: a name of LARGE_CSG_CASES or its index"""
    if not isinstance(case, str):
        case = LARGE_CSG_CASES[int(case)]
    rng = np.random.default_rng({"groups_of_spheres": 1, "many_cylinders": 2, "mesh_minus_sphere": 3, "glass_clusters": 4}[case])
    _SPREAD[0] = 0.9

    def cluster(n, make):
        return S.Group(S.Material(), _identity(), [make() for _ in range(n)])

    def ball(mat=None):
        return S.Sphere(mat or _material(rng), translation(*rng.uniform(-0.9, 0.9, 3)) * scaling(*rng.uniform(0.3, 0.6, 3)))
    if case == "groups_of_spheres":
        obj = S.CSG(S.Material(), rotation_y(0.4), S.CSGDifference, cluster(10, ball), cluster(10, ball))
    elif case == "many_cylinders":
        def rod():
            cls = S.Cylinder if rng.random() < 0.6 else S.Cone
            return cls(_material(rng), translation(*rng.uniform(-0.8, 0.8, 3)) * rotation_x(rng.uniform(0, 3)) * rotation_z(rng.uniform(0, 3)) *
                       scaling(0.25, 0.8, 0.25), minimum=-1.0, maximum=1.0, closed=True)
        obj = S.CSG(S.Material(), _identity(), S.CSGUnion, S.CSG(S.Material(), _identity(), S.CSGIntersect, cluster(6, rod), S.Sphere(_material(rng), scaling(1.3, 1.3, 1.3))),
                    S.CSG(S.Material(), translation(0.3, 0, 0), S.CSGDifference, cluster(5, rod), S.Cube(_material(rng), scaling(0.5, 0.5, 0.5))))
    elif case == "mesh_minus_sphere":
        obj = S.CSG(S.Material(), rotation_x(0.3), S.CSGDifference, _mesh(rng, 7, 10, smooth=True),
                    S.Sphere(_glass(rng), translation(0.5, 0.4, -0.6) * scaling(0.8, 0.8, 0.8)))
    else:
        obj = S.CSG(S.Material(), _identity(), S.CSGUnion, cluster(9, lambda: ball(_glass(rng))), cluster(9, lambda: ball(_glass(rng))))
    floor = S.Plane(_material(rng), translation(0, -2.2, 0))
    floor.material.transparency = np.float64(0.0)
    lights = [point_light(point(-4, 6, -5), _rgb(rng, 0.6, 1.0)), point_light(point(5, 3, -4), _rgb(rng, 0.2, 0.5))]
    cam = Camera(HSIZE, VSIZE, 0.8, view_transform(point(0.4, 1.6, -5.5), point(0, 0, 0), vector(0, 1, 0)))
    world = World(lights, [obj, floor, ball()])
    return cam, world


@pytest.mark.parametrize("case", LARGE_CSG_CASES)
def test_csg_larger_than_the_hit_list(case, oracle_cls, monkeypatch):
    """A CSG object whose leaves could put more than 32 intersections on one ray -- two groups of ten spheres, a dozen
    capped cylinders and cones, a tessellated mesh cut by a sphere, overlapping glass clusters (the refraction container walk
    inside the CSG) -- used to be refused at rtc_scene_create; the reference renders all of them (CSG.local_intersect over
    arbitrary groups, shapes.py:366-388).  Such roots are evaluated without a hit list (csg_stream): against the oracle.
    Then every CSG of an ordinary scene through the same code (RTC_CSG_STREAM_ALL): the same frame bit for bit."""
    cam, world = large_csg_world(case)
    cs = compile_scene(world)
    assert sum({0: 2, 1: 1, 2: 2, 3: 4, 4: 4, 5: 1}[int(k)] for k in cs.arrays["leaf_kind"][:cs.counts["n_leaves"] - 2]) > 32
    xs, ys = np.meshgrid(np.arange(HSIZE), np.arange(VSIZE), indexing="ij")
    orc = oracle_cls(world)
    want, leaf, t_hit, rays = orc.render_pixels(cam, xs.ravel(), ys.ravel())
    want, leaf = want.reshape(HSIZE, VSIZE, 3), leaf.reshape(HSIZE, VSIZE)
    with _ffi.Scene(cs, device=0) as sc:
        st = {}
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
        f32, _ = sc.render(cam, precision="fp32")
    _exact("large csg " + case, orc, cam, rgb, want, ids, leaf)
    assert [st["rays"][0], st["rays"][1] + st["shadow_rays_culled"], st["rays"][2], st["rays"][3]] == [int(v) for v in rays]
    assert frac_off_by_more_than_one(f32, want) <= 0.005
    assert (ids >= 0).mean() > 0.2 and len(np.unique(ids)) > 6


@pytest.mark.parametrize("seed", range(6))
def test_streamed_csg_equals_listed_csg(seed, monkeypatch):
    """The two evaluations of a CSG root -- per-lane hit list, and list-free enumeration (csg_stream) -- on the nested CSG
    worlds of test_random_csg_world (glass and mirror parts included): bit-identical frames, ids and ray counts."""
    rng = np.random.default_rng(5000 + seed)
    _SPREAD[0] = 1.0
    objs = [_csg_tree(rng, 2, translation(*rng.uniform(-1.6, 1.6, 3)) * rotation_x(rng.uniform(0, 6.28))) for _ in range(int(rng.integers(1, 4)))]
    floor = S.Plane(_material(rng), translation(0, -2.8, 0))
    floor.material.transparency = np.float64(0.0)
    world = World([point_light(point(-3, 7, -5), _rgb(rng, 0.5, 1.0))], objs + [floor])
    cam = Camera(HSIZE, VSIZE, 0.9, view_transform(point(0.3, 2.5, -6.5), point(0, -0.2, 0), vector(0, 1, 0)))
    cs = compile_scene(world)
    out = []
    for stream_all in (False, True):
        if stream_all:
            monkeypatch.setenv("RTC_CSG_STREAM_ALL", "1")
        with _ffi.Scene(cs, device=0) as sc:
            st = {}
            rgb, ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
            out.append((rgb, ids, st["rays"]))
    monkeypatch.delenv("RTC_CSG_STREAM_ALL")
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2]


@pytest.mark.parametrize("seed", [4, 8, 11, 18, 20, 22])          # worlds with many reflection / refraction rays
def test_random_world_at_other_depths(seed, oracle_cls):
    """`remaining` other than the reference's default 5 (renderer.py:562): 0 (no secondary rays), 1, 3 and the library's
    maximum 6, against the oracle run with the same `remaining`."""
    cam, world = random_world(seed)
    xs, ys = np.meshgrid(np.arange(HSIZE), np.arange(VSIZE), indexing="ij")
    orc = oracle_cls(world)
    with _ffi.Scene(compile_scene(world), device=0) as sc:
        for depth in (0, 1, 3, 6):
            want, leaf, _, rays = orc.render_pixels(cam, xs.ravel(), ys.ravel(), remaining=depth)
            want = want.reshape(HSIZE, VSIZE, 3)
            st = {}
            rgb, _ = sc.render(cam, precision="fp64", max_depth=depth, stats=st)
            _exact("seed %d depth %d" % (seed, depth), orc, cam, rgb, want, remaining=depth)
            if depth == 0:
                assert st["rays"][2] == 0 and st["rays"][3] == 0
        with pytest.raises(_ffi.RtcError, match="max_depth"):
            sc.render(cam, precision="fp64", max_depth=7)


def random_texture_world(seed):
    rng = np.random.default_rng(9000 + seed)
    _SPREAD[0] = 1.4
    objs = []
    for k in range(int(rng.integers(2, 6))):
        o = _primitive(rng)
        r = rng.random()
        if r < 0.3:
            pat = S.TextureMapPattern(_uv_pattern(rng, True), [S.PlaneUVMap, S.SphereUVMap][rng.integers(2)], _small_transform(rng))
        elif r < 0.6:
            pat = S.CylinderMapPattern(_uv_pattern(rng, True), _uv_pattern(rng, True), S.CylinderUVMap, _small_transform(rng))
        else:
            pat = S.CubeMapPattern(*[_uv_pattern(rng, True) for _ in range(6)], S.CubeUVMap, _small_transform(rng))
        o.material.pattern = pat
        o.material.transparency = np.float64(0.0)
        o.material.reflective = np.float64(0.0)            # the subject is texel lookup; a mirror cone around the camera is chaos
        objs.append(o)
    floor = S.Plane(_material(rng), translation(0, -2.4, 0))
    floor.material.pattern = S.TextureMapPattern(_uv_pattern(rng, True), S.PlaneUVMap, scaling(*rng.uniform(0.5, 3, 3)))
    floor.material.transparency = np.float64(0.0)
    objs.append(floor)
    lights = [point_light(point(*(rng.uniform(-5, 5, 3) + np.array([0, 7, -5]))), _rgb(rng, 0.6, 1.0))]
    cam = Camera(HSIZE, VSIZE, 0.9, view_transform(point(0.4, 3.0, -6.0), point(0, -0.4, 0), vector(0, 1, 0)))
    world = World(lights, objs)
    return cam, world


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RTC_FUZZ_SEEDS", "12"))))
def test_random_image_textures(seed, oracle_cls):
    """Image textures (earth, skybox) with random texels and sizes under the planar, spherical, cylindrical and cube maps,
    on the primitive each map is meant for and on others, rotated and scaled: texel lookup (np.round half-to-even, the
    image maximum as the scale) against the oracle."""
    cam, world = random_texture_world(seed)
    xs, ys = np.meshgrid(np.arange(HSIZE), np.arange(VSIZE), indexing="ij")
    orc = oracle_cls(world)
    want, leaf, t_hit, _ = orc.render_pixels(cam, xs.ravel(), ys.ravel())
    want, leaf = want.reshape(HSIZE, VSIZE, 3), leaf.reshape(HSIZE, VSIZE)
    near = (~(np.isfinite(t_hit) & (t_hit > 60.0))).reshape(HSIZE, VSIZE)
    with _ffi.Scene(compile_scene(world), device=0) as sc:
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
        f32, _ = sc.render(cam, precision="fp32")
    ok = np.isfinite(want).all(axis=2)                                    # the reference raises IndexError outside an image; the oracle says NaN
    off = (np.abs(rgb - want).max(axis=2) > 1e-9) & ok
    print("tex seed %d: fp64 off %d, ids %d, nan px %d, fp32 off>1 %.4f" % (seed, off.sum(), (ids != leaf).sum(), (~ok).sum(),
          frac_off_by_more_than_one(f32[near & ok], want[near & ok])))
    _exact("tex seed %d" % seed, orc, cam, rgb, want, ids, leaf, ok & np.isfinite(rgb).all(axis=2))
    assert frac_off_by_more_than_one(f32[near & ok], want[near & ok]) <= 0.005     # 400 texture worlds: 99 % within 0.001, worst 0.0036 (texel boundaries are everywhere at 64 x 48)


@pytest.mark.parametrize("distance", [1.0e3, 1.0e4, 1.0e5])
def test_camera_far_from_small_geometry(distance, oracle_cls):
    """The box tests in front of the leaf tests run in fp32 whatever the precision; fp32 arithmetic relative to a ray
    origin thousands of units away would displace the boxes of unit-sized geometry by more than their padding and could
    cull hits the leaf test accepts (silently, in fp64 mode too).  Rays are re-based where they enter each tree: a
    telephoto camera at 1e3, 1e4 and 1e5 units from a unit-sized mesh among a few primitives, off-axis so that no
    direction component is small by accident, against the oracle: hit ids exact, colours to 1e-6 * max(1, distance / 1e4)^2
    (seen from D units the reference's own sphere / cylinder discriminants b^2 - 4ac cancel 2 log10(D) digits, so a hit
    distance carries an absolute error of about 1e-16 D^2 that depends on whether the compiler fused a multiply-add: 1e-8
    at 1e4, 1e-6 at 1e5; a culled hit would change an id, and a colour by far more).  At 1e5 the oracle's own shading of
    the small sphere and cylinder is noise near their terminators (the discriminant's error divided by a small sqrt reaches
    the 1e-4 shadow offset: 2.6 % of the pixels flip between lit and self-shadowed under a 1e-9 rad turn of the camera, in
    the oracle alone), so the share of explained differences allowed there is 3 %, not 1 %; ids stay exact everywhere."""
    rng = np.random.default_rng(77)
    _SPREAD[0] = 0.8
    objs = [_mesh(rng, 9, 14, smooth=True), S.Sphere(_material(rng), translation(1.4, 0.2, 0.3) * scaling(0.4, 0.4, 0.4)),
            S.Cube(_material(rng), translation(-1.3, -0.3, 0.2) * rotation_y(0.5) * scaling(0.35, 0.35, 0.35)),
            S.Cylinder(_material(rng), translation(0.2, -0.9, -0.8) * scaling(0.3, 0.3, 0.3), minimum=-1.0, maximum=1.0, closed=True)]
    for o in objs[1:]:
        o.material.pattern = None
    lights = [point_light(point(-4, 6, -5), _rgb(rng, 0.6, 1.0))]
    eye = np.array([0.48, 0.31, -0.82]) / np.linalg.norm([0.48, 0.31, -0.82]) * distance
    fov = 2 * np.arctan(2.4 / distance)
    cam = Camera(96, 64, fov, view_transform(point(*eye), point(0.05, -0.1, 0), vector(0, 1, 0)))
    world = World(lights, objs)
    xs, ys = np.meshgrid(np.arange(cam.hsize), np.arange(cam.vsize), indexing="ij")
    orc = oracle_cls(world)
    want, leaf, _, rays = orc.render_pixels(cam, xs.ravel(), ys.ravel())
    want, leaf = want.reshape(cam.hsize, cam.vsize, 3), leaf.reshape(cam.hsize, cam.vsize)
    assert (leaf >= 0).mean() > 0.1 and len(np.unique(leaf)) > 50
    with _ffi.Scene(compile_scene(world), device=0) as sc:
        rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
        sc.build_bvh_on_device()                                  # the GPU-built tree re-bases on its own root box
        rgb2, ids2 = sc.render(cam, precision="fp64", want_ids=True)
    tol, allowed = 1e-6 * max(1.0, distance / 1e4) ** 2, 0.01 if distance < 1e5 else 0.03
    assert (ids == leaf).all() and (ids2 == leaf).all()
    _exact("camera at %g" % distance, orc, cam, rgb, want, ids, leaf, allowed=allowed, tol=tol)
    _exact("camera at %g, GPU-built tree" % distance, orc, cam, rgb2, want, ids2, leaf, allowed=allowed, tol=tol)


# ---------------------------------------------------------------------------------------------------------------
# instanced meshes (rtc_scene_desc::inst, compile_scene(instancing=True)): loader-made meshes placed several times
# ---------------------------------------------------------------------------------------------------------------
def _ball_obj(rng, n_lat, n_lon, smooth):
    """OBJ text of a tessellated ellipsoid (two poles, n_lat - 1 rings of n_lon vertices), with the ellipsoid's normals at the
    vertices if `smooth`, else every face's own normal at its three corners"""
    radii = rng.uniform(0.5, 1.2, 3)
    pts = [np.array([0.0, radii[1], 0.0])]
    for i in range(1, n_lat):
        for j in range(n_lon):
            th, ph = np.pi * i / n_lat, 2 * np.pi * j / n_lon
            pts.append(radii * np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)]))
    pts.append(np.array([0.0, -radii[1], 0.0]))
    lines = ["v %.12f %.12f %.12f" % tuple(p) for p in pts]
    if smooth:
        for p in pts:
            n = p / (radii * radii)
            lines.append("vn %.12f %.12f %.12f" % tuple(n / np.linalg.norm(n)))
    ring = lambda i, j: 2 + (i - 1) * n_lon + (j % n_lon)          # 1-based OBJ index of ring i (1 .. n_lat-1), column j
    last = len(pts)
    faces = [(1, ring(1, j + 1), ring(1, j)) for j in range(n_lon)]
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j + 1), ring(i + 1, j)
            faces += [(a, b, c), (a, c, d)]
    faces += [(last, ring(n_lat - 1, j), ring(n_lat - 1, j + 1)) for j in range(n_lon)]
    if not smooth:                                                 # (the loader, like the reference's, wants a normal per corner: the face's own)
        for a, b, c in faces:
            n = np.cross(pts[b - 1] - pts[a - 1], pts[c - 1] - pts[a - 1])
            lines.append("vn %.12f %.12f %.12f" % tuple(n / np.linalg.norm(n)))
    lines.append("g ball")
    for fi, f in enumerate(faces):
        lines.append("f " + " ".join("%d//%d" % (k, k if smooth else fi + 1) for k in f))
    return "\n".join(lines), len(faces)


def _has_perturbed(p):
    if p is None:
        return False
    if type(p).__name__ == "PerturbedPattern":
        return True
    return any(_has_perturbed(getattr(p, k, None)) for k in ("p1", "p2", "p3"))


def _plain_material(rng):
    """_material without perturbed patterns (a scene that has one is rendered flattened: its kernels carry no instance walk)"""
    m = _material(rng)
    if _has_perturbed(m.pattern):
        m.pattern = _simple_pattern(rng)
    return m


def random_instanced_world(seed):
    """One or two loader-made meshes, two to four instances each (at most eight together) under random transforms -- shear
    and non-uniform scale among them, some inside a group with a transform of its own -- with random materials (glass, mirrors,
    patterns in the instance's object space, shadow-less ones), a few primitives and a floor."""
    import io
    from ray_tracer_challenge_amd.obj_parser import OBJParser, obj_to_group
    rng = np.random.default_rng(9000 + seed)
    _SPREAD[0] = 1.6
    objs = []
    for _ in range(int(rng.integers(1, 3))):
        text, n = _ball_obj(rng, int(rng.integers(5, 9)), int(rng.integers(7, 12)), smooth=bool(rng.random() < 0.6))
        for _ in range(int(rng.integers(2, 5))):
            g = obj_to_group(OBJParser(io.StringIO(text), default_material=_plain_material(rng)))
            g.transform = _transform(rng)
            objs.append(S.Group(S.Material(), _small_transform(rng), [g]) if rng.random() < 0.3 else g)
    for _ in range(int(rng.integers(0, 4))):
        p = _primitive(rng)
        if _has_perturbed(p.material.pattern):
            p.material.pattern = None
        objs.append(p)
    floor = S.Plane(_plain_material(rng), translation(0, -2.6, 0))
    floor.material.transparency = np.float64(0.0)
    objs.append(floor)
    lights = [point_light(point(*(rng.uniform(-5, 5, 3) + np.array([0, 7, -5]))), _rgb(rng, 0.5, 1.0))]
    if seed % 3 == 0:
        lights.append(point_light(point(*(rng.uniform(-5, 5, 3) + np.array([0, 4, -6]))), _rgb(rng, 0.2, 0.6)))
    cam = Camera(71, 45, 0.9, view_transform(point(0.5, 1.8, -7.0), point(0, 0, 0), vector(0, 1, 0)))
    return cam, World(lights, objs)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RTC_FUZZ_INST_SEEDS", "6"))))
def test_random_instanced_world(seed, oracle_cls, monkeypatch):
    """Repeated meshes as instances (one object-space tree per mesh, the ray transformed per instance) and flattened to world
    space: both against the oracle, which walks the world's own groups."""
    from ray_tracer_challenge_amd import compile as compile_mod
    monkeypatch.setattr(compile_mod, "MIN_INSTANCE_TRIS", 32)          # (the product's threshold is 1024 triangles: small meshes keep the oracle quick)
    cam, world = random_instanced_world(seed)
    w, h = cam.hsize, cam.vsize
    xs, ys = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    orc = oracle_cls(world)
    want, leaf, t_hit, _ = orc.render_pixels(cam, xs.ravel(), ys.ravel())
    want, leaf = want.reshape(w, h, 3), leaf.reshape(w, h)
    near = (~(np.isfinite(t_hit) & (t_hit > 60.0))).reshape(w, h)
    inst, flat = compile_scene(world, instancing=True), compile_scene(world, instancing=False)
    assert 2 <= inst.counts["n_instances"] <= 8 and flat.counts["n_instances"] == 0
    assert inst.counts["n_tris"] < flat.counts["n_tris"]
    for label, cs in (("instanced", inst), ("flattened", flat)):
        with _ffi.Scene(cs, device=0) as sc:
            rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
            f32, _ = sc.render(cam, precision="fp32")
        ok = np.isfinite(want).all(axis=2) & np.isfinite(rgb).all(axis=2)
        assert np.array_equal(np.isfinite(want).all(axis=2), np.isfinite(rgb).all(axis=2))
        _exact("instanced seed %d, %s" % (seed, label), orc, cam, rgb, want, ids, leaf, ok)
        assert frac_off_by_more_than_one(f32[ok & near], want[ok & near]) <= 0.01
