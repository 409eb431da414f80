"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the fixtures the
reference itself rendered.  fp64: colours to 1e-9 and exact hit-leaf ids on EVERY pixel that is not
a tie, a grazing hit, acne or on an edge as SURVEY.md 8d defines them -- each differing pixel is
classified from the oracle's own hit list (conftest.explain_mismatches) and none may stay
unexplained.  fp32: the north_star gate -- every channel within 1/255 after PPM quantisation on
>= 99.9 % of pixels."""
import ctypes as C

import numpy as np
import pytest

from conftest import explain_mismatches, frac_off_by_more_than_one, load_golden, load_pack, move_triangles, ppm_levels, record_classes
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.canvas import quantize
from ray_tracer_challenge_amd.compile import compile_scene

pytestmark = pytest.mark.gpu

# (pack, render size or None = as shipped, oracle pixel stride)
CASES = [("cover", None, 1), ("csg_test", (192, 192), 1), ("group", None, 1), ("checkered_sphere", (160, 160), 1),
         ("checkered_plane", (160, 160), 1), ("checkered_cylinder", (160, 160), 1), ("align_check_plane", (160, 160), 1),
         ("checkered_cube", (320, 160), 1), ("earth", (320, 160), 1), ("skybox", (320, 160), 1),
         ("shadow_glamour_shot_nojitter", (240, 96), 1), ("teapot", None, 3), ("dragons", (480, 192), 4),
         ("perturbed", (320, 200), 1),          # perturbed patterns: noise restated from the published algorithm, parity unpinned
         ("patterns", None, 1),                 # every solid / composite pattern no reference scene uses (reference-rendered fixture)
         ("uvmaps", None, 1)]                   # UV maps with pattern transforms, on other shapes than their namesakes (likewise)
GOLDEN = [c[0] for c in CASES]
_ORACLE_SAMPLES = {}


@pytest.fixture(scope="module")
def scenes():
    cache = {}

    def get(name):
        if name not in cache:
            cam, world, _ = load_pack(name)
            cache[name] = (cam, world, _ffi.Scene(compile_scene(world), device=0))
        return cache[name]
    yield get
    for _, _, sc in cache.values():
        sc.close()


def _queries(st):
    """Ray counts as the reference traces them: is_shadowed queries answered without a ray (light behind the surface: the
    result is the ambient term either way) count as the shadow rays the reference spends on them."""
    r = list(st["rays"])
    r[1] += st["shadow_rays_culled"]
    return r


def _sample(cam, stride):
    xs, ys = np.meshgrid(np.arange(stride // 2, cam.hsize, stride), np.arange(stride // 2, cam.vsize, stride), indexing="ij")
    return xs.ravel(), ys.ravel()


@pytest.mark.parametrize("name,size,stride", CASES)
def test_fp64_matches_oracle(name, size, stride, scenes, oracle_cls):
    cam, world, sc = scenes(name)
    if size:
        cam = cam.resized(*size)
    st = {}
    rgb, ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
    px, py = _sample(cam, stride)
    orc = oracle_cls(world)
    want, leaf, t, rays = orc.render_pixels(cam, px, py)
    _ORACLE_SAMPLES[(name, cam.hsize, cam.vsize, stride)] = (orc, px, py, want, leaf)      # test_triangle_bvh_built_on_the_gpu uses them again
    unexplained, classes = explain_mismatches(orc, cam, px, py, rgb[px, py], want, ids[px, py], leaf)
    print("%s: %d pixels, differing and explained: %s" % (name, len(px), classes))
    record_classes("fp64 vs oracle: %s %dx%d stride %d" % (name, cam.hsize, cam.vsize, stride), len(px), classes, len(unexplained))
    assert not unexplained, unexplained[:20]
    assert sum(classes.values()) <= 0.002 * len(px) + 2
    assert st["rays"][0] == cam.hsize * cam.vsize
    if stride == 1:
        assert _queries(st) == [int(v) for v in rays]          # same rays, kind by kind
        assert st["shadow_rays_culled"] >= 0


@pytest.mark.parametrize("name", GOLDEN)
def test_fp64_matches_reference_fixture(name, scenes, oracle_cls):
    g = load_golden(name)
    cam, world, sc = scenes(name)
    cam = cam.resized(g["meta"]["hsize"], g["meta"]["vsize"])
    rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
    got, got_ids = rgb[g["px"], g["py"]], ids[g["px"], g["py"]]
    # the fixture is the reference's own output; a pixel that differs must be one the oracle (which agrees with the
    # fixture, tests/test_oracle_golden.py) shows to be a tie / grazing / acne / edge pixel
    unexplained, classes = explain_mismatches(oracle_cls(world), cam, g["px"], g["py"], got, g["rgb"], got_ids, g["hit_leaf"])
    record_classes("fp64 vs reference fixture: %s %dx%d" % (name, cam.hsize, cam.vsize), len(g["px"]), classes, len(unexplained))
    assert not unexplained, unexplained[:20]
    assert sum(classes.values()) <= 0.002 * len(g["px"]) + 1
    assert frac_off_by_more_than_one(got, g["rgb"]) <= 0.0005


@pytest.mark.parametrize("name,size,stride", CASES)
def test_fp32_gate(name, size, stride, scenes, oracle_cls):
    cam, world, sc = scenes(name)
    if size:
        cam = cam.resized(*size)
    rgb, ids = sc.render(cam, precision="fp32", want_ids=True)
    px, py = _sample(cam, stride)
    want, leaf, t, _ = oracle_cls(world).render_pixels(cam, px, py)
    assert frac_off_by_more_than_one(rgb[px, py], want) <= 0.001
    # hit ids: exact outside grazing / acne / tie pixels; those are a small fraction of any of these frames
    assert (ids[px, py] != leaf).mean() <= 0.01


def test_fp32_matches_reference_fixtures(scenes):
    for name in GOLDEN:
        g = load_golden(name)
        cam, world, sc = scenes(name)
        cam = cam.resized(g["meta"]["hsize"], g["meta"]["vsize"])
        rgb, _ = sc.render(cam, precision="fp32")
        assert frac_off_by_more_than_one(rgb[g["px"], g["py"]], g["rgb"]) <= 0.001, name


# ---------------------------------------------------------------- size-independent properties at full size
def test_full_size_properties(scenes):
    cam, world, sc = scenes("dragons")
    cam = cam.resized(1920, 1080)
    st = {}
    full, ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
    n_lights = 4
    shaded_primary = int((ids >= 0).sum())
    assert st["rays"][0] == 1920 * 1080
    assert _queries(st)[1] % n_lights == 0 and _queries(st)[1] >= n_lights * shaded_primary
    assert 0 < st["shadow_rays_culled"] < _queries(st)[1]                      # closed meshes lit from four sides
    assert np.all(full[ids < 0] == 0) and np.isfinite(full).all()
    # idempotence / determinism
    again, _ = sc.render(cam, precision="fp64")
    np.testing.assert_array_equal(full, again)
    # tiles == whole frame, bit for bit, in any order
    tx, ty = _ffi.tiles_for(cam.hsize, cam.vsize)
    tiles = np.random.default_rng(0).permutation(tx * ty).astype(np.int32)[:300]
    st2 = {}
    rt, it = sc.render_tiles(cam, tiles, precision="fp64", want_ids=True, stats=st2)
    out = np.full_like(full, np.nan)
    oid = np.full_like(ids, -7)
    _ffi.scatter_tiles(out, oid, tiles, rt, it)
    m = ~np.isnan(out[:, :, 0])
    assert m.sum() > 0.1 * m.size
    np.testing.assert_array_equal(out[m], full[m])
    np.testing.assert_array_equal(oid[m], ids[m])
    # depth 0: no secondary rays; deeper never traces fewer
    st0, st1 = {}, {}
    sc.render(cam, precision="fp64", max_depth=0, stats=st0)
    sc.render(cam, precision="fp64", max_depth=1, stats=st1)
    assert st0["rays"][2:] == [0, 0] and _queries(st0)[1] == n_lights * shaded_primary
    assert st1["rays"][2] <= st["rays"][2] and st1["rays"][3] <= st["rays"][3] and st1["rays"][2] > 0


def test_benched_frame_against_the_oracle(scenes, oracle_cls):
    """The frame bench.py times -- dragons at 1920x1080, depth 5 -- on an even grid of 2 294 of its pixels against the
    oracle (brute force over 140 940 triangles): colours, hit ids and every differing pixel classified."""
    cam, world, sc = scenes("dragons")
    cam = cam.resized(1920, 1080)
    rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
    px, py = _sample(cam, 30)
    orc = oracle_cls(world)
    want, leaf, t, _ = orc.render_pixels(cam, px, py)
    assert len(px) >= 2000 and (leaf >= 0).sum() > 300
    unexplained, classes = explain_mismatches(orc, cam, px, py, rgb[px, py], want, ids[px, py], leaf)
    print("dragons 1080p: %d pixels, differing and explained: %s" % (len(px), classes))
    record_classes("fp64 vs oracle: dragons 1920x1080 (the benched frame) stride 30", len(px), classes, len(unexplained))
    assert not unexplained, unexplained[:20]
    assert sum(classes.values()) <= 0.002 * len(px) + 2
    rgb32, ids32 = sc.render(cam, precision="fp32", want_ids=True)
    assert frac_off_by_more_than_one(rgb32[px, py], want) <= 0.001


def test_instrumented_variant_is_identical(scenes):
    cam, world, sc = scenes("cover")
    a, _ = sc.render(cam, precision="fp64")
    st = {}
    b, _ = sc.render(cam, precision="fp64", stats=st, count_work=True)
    np.testing.assert_array_equal(a, b)
    assert st["work"][4] > 0 and st["work"][0] > 0 and st["work"][2] > 0


def test_device_pointer_path_and_quantize_kernel(scenes):
    import torch
    cam, world, sc = scenes("cover")
    host, _ = sc.render(cam, precision="fp64")
    dev = torch.device("cuda", 0)
    frame = torch.zeros((cam.hsize, cam.vsize, 3), dtype=torch.float64, device=dev)
    rays = torch.zeros(13, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    sc.render_device(cam, frame.data_ptr(), rays_ptr=rays.data_ptr(), precision="fp64", stream=stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(frame.cpu().numpy(), host)
    assert int(rays[0]) == cam.hsize * cam.vsize
    for clamp in (True, False):
        maxc = 1.0 if clamp else float(frame.max())
        out = torch.zeros((cam.vsize, cam.hsize, 3), dtype=torch.uint8, device=dev)
        rc = _ffi.lib().rtc_quantize_device(C.c_void_p(frame.data_ptr()), cam.hsize, cam.vsize, maxc, C.c_void_p(out.data_ptr()),
                                            C.c_void_p(stream))
        assert rc == 0
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), quantize(host, clamp))


@pytest.mark.parametrize("name,size", [("teapot", (480, 270)), ("csg_test", (192, 192))])
def test_device_path_accumulates_ray_counts(name, size, scenes):
    """`rays_dev` is added to, never cleared, by `rtc_render_device` (bench.py zeroes it once per timed loop): after k launches
    it holds k frames' counts -- through the per-wave slots + sum_rays_kernel (scenes without CSG) and through the
    workgroups' atomics (CSG scenes; and launches without feedback) alike, in every stage of the launch-order feedback."""
    import torch
    cam, world, sc = scenes(name)
    cam = cam.resized(*size)
    one = {}
    sc.render(cam, precision="fp64", stats=one, feedback=False)
    dev = torch.device("cuda", 0)
    frame = torch.zeros((cam.hsize, cam.vsize, 3), dtype=torch.float64, device=dev)
    rays = torch.zeros(13, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for k in range(1, 9):
        sc.render_device(cam, frame.data_ptr(), rays_ptr=rays.data_ptr(), precision="fp64", stream=stream)
        torch.cuda.synchronize()
        got = [int(v) for v in rays.tolist()]
        assert got[:4] == [k * v for v in one["rays"]], k
        assert got[12] == k * one["shadow_rays_culled"], k
        assert got[4:12] == [0] * 8


def test_render_api_and_multi(scenes):
    from ray_tracer_challenge_amd.canvas import construct_ppm
    from ray_tracer_challenge_amd.renderer import device_scenes, render, render_multi
    cam, world, sc = scenes("cover")
    image = render(cam, world)
    arr = render_multi(cam, world, num_threads=4)
    assert arr.shape == (cam.hsize, cam.vsize, 3) and arr.dtype == np.float64
    np.testing.assert_array_equal(image.np_arr, arr)
    with device_scenes(world, 1) as resident:                    # many frames of one world: compiled and uploaded once
        for size in ((100, 100), (64, 48)):
            c2 = cam.resized(*size)
            got, ids = render_multi(c2, None, scenes=resident, return_ids=True)
            ref, ref_ids = sc.render(c2, precision="fp64", want_ids=True)
            np.testing.assert_array_equal(got, ref)
            np.testing.assert_array_equal(ids, ref_ids)
    assert len(construct_ppm(image)) == len("P6\n100 100\n255\n") + 3 * 100 * 100 + 1


# ---------------------------------------------------------------- edge cases
def test_edge_sizes_and_empty_worlds(scenes, oracle_cls):
    from ray_tracer_challenge_amd import shapes as S
    from ray_tracer_challenge_amd.renderer import World, point_light
    from ray_tracer_challenge_amd.canvas import color
    from ray_tracer_challenge_amd.vector import point
    cam, world, sc = scenes("cover")
    for size in [(1, 1), (7, 3), (33, 17), (8, 64)]:           # the reference's render_multi fails on odd sizes; render does not
        c = cam.resized(*size)
        got, ids = sc.render(c, precision="fp64", want_ids=True)
        want, leaf, _ = oracle_cls(world).render(c)
        np.testing.assert_allclose(got, want, atol=1e-9)
        np.testing.assert_array_equal(ids, leaf)
    empty = World([point_light(point(0, 0, 0), color(1, 1, 1))], [])
    with _ffi.Scene(compile_scene(empty)) as s0:
        got, ids = s0.render(cam.resized(16, 16), want_ids=True)
    assert np.all(got == 0) and np.all(ids == -1)
    dark = World([], [S.sphere()])
    with _ffi.Scene(compile_scene(dark)) as s1:
        got, ids = s1.render(cam.resized(16, 16), want_ids=True)
    assert np.all(got == 0)
    with pytest.raises(_ffi.RtcError):
        sc.render(cam, max_depth=7)
    with pytest.raises(_ffi.RtcError):
        sc.render_tiles(cam, [10 ** 6])
    rt, _ = sc.render_tiles(cam, [])
    assert rt.shape[0] == 0


def test_nested_and_blended_patterns_and_nested_csg(oracle_cls):
    from ray_tracer_challenge_amd import shapes as S
    from ray_tracer_challenge_amd.canvas import color
    from ray_tracer_challenge_amd.matrix import rotation_y, scaling, translation
    from ray_tracer_challenge_amd.renderer import Camera, World, point_light, view_transform
    from ray_tracer_challenge_amd.vector import point, vector
    W, B, Rd, G = color(1, 1, 1), color(0, 0, 0), color(1, 0, 0), color(0, 1, 0)
    floor = S.plane()
    floor.transform = translation(0, -1, 0)
    floor.material.pattern = S.NestedPattern(S.Checkers(W, B, scaling(2, 2, 2)), S.Stripe(Rd, W, scaling(0.2, 1, 1)),
                                             S.BlendedPattern(S.Ring(G, B, scaling(0.3, 1, 0.3)), S.Gradient(W, Rd)))
    ball = S.sphere()
    ball.material.pattern = S.RadialGradient(Rd, G, scaling(0.25, 0.25, 0.25))
    cone = S.Cone(S.Material(color=color(0.3, 0.6, 1), reflective=0.3), translation(2.2, 0, 0.5), -1, 0, True)
    cyl = S.Cylinder(S.Material(color=color(1, 0.8, 0.2), transparency=0.6, refractive_index=1.3), translation(-2.2, -1, 0), 0, 1.5, True)
    inner = S.CSG(S.Material(), translation(0, 0, 0), S.CSGIntersect, S.Cube(S.Material(color=Rd), scaling(0.8, 0.8, 0.8)),
                  S.Sphere(S.Material(color=G), None))
    lens = S.CSG(S.Material(), translation(0, 0.2, -2.2) * rotation_y(0.5) * scaling(0.6, 0.6, 0.6), S.CSGDifference, inner,
                 S.Cylinder(S.Material(color=W, transparency=0.5, reflective=0.2, refractive_index=1.5), scaling(0.4, 1, 0.4), -2, 2, True))
    union = S.CSG(S.Material(), translation(0, 1.6, 0), S.CSGUnion, S.Sphere(S.Material(color=Rd), scaling(0.5, 0.5, 0.5)),
                  S.Sphere(S.Material(color=B + 0.5), translation(0.4, 0, 0) * scaling(0.5, 0.5, 0.5)))
    world = World([point_light(point(-5, 6, -6), color(1, 1, 1)), point_light(point(4, 3, -5), color(0.3, 0.3, 0.3))],
                  [S.Group(S.Material(), None, [floor, ball, cone, cyl, lens, union])])
    cam = Camera(200, 120, 0.9, view_transform(point(0, 2.5, -7), point(0, 0.2, 0), vector(0, 1, 0)))
    want, leaf, rays = oracle_cls(world).render(cam)
    with _ffi.Scene(compile_scene(world)) as sc:
        st = {}
        got, ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
        got32, _ = sc.render(cam, precision="fp32")
    xs, ys = np.meshgrid(np.arange(cam.hsize), np.arange(cam.vsize), indexing="ij")
    unexplained, classes = explain_mismatches(oracle_cls(world), cam, xs.ravel(), ys.ravel(), got.reshape(-1, 3), want.reshape(-1, 3), ids.ravel(), leaf.ravel())
    assert not unexplained, (unexplained[:20], classes)
    assert sum(classes.values()) <= 0.002 * xs.size
    assert _queries(st) == [int(v) for v in rays]
    # fine stripes / rings / checkers on a plane seen at a grazing angle: fp32 flips the pattern cell on
    # boundary pixels, which is inherent to the precision; this is a torture scene, not a BASELINE config
    assert frac_off_by_more_than_one(got32, want) <= 0.02


@pytest.mark.parametrize("name,size", [("cover", None), ("dragons", (480, 192)), ("teapot", None), ("shadow_glamour_shot", (240, 96)),
                                       ("shadow_glamour_shot_nojitter", (240, 96)), ("csg_test", (192, 192)), ("perturbed", (160, 100))])
def test_launch_order_feedback_changes_nothing(name, size, scenes):
    """Frames after the first launch the dearest 8x8 blocks first (costs recorded by the previous frame on the
    same scene handle), and from the third frame on with the dearest blocks split into 4x4 or 2x2 parts whose pixels
    are each traced by 4 or 16 lanes that share the shadow rays out among themselves (point lights: one light each;
    the area light's 100 cells: 25 or 7 rounds), and fp64 frames are timed at two register budgets (kernel variants for 3
    and 2 waves per SIMD): any launch order, any split and either variant must give bit-identical images, ids and ray counts."""
    cam, world, sc = scenes(name)
    if size:
        cam = cam.resized(*size)
    for prec in ("fp64", "fp32"):
        s0 = {}
        plain, ids0 = sc.render(cam, precision=prec, want_ids=True, stats=s0, feedback=False)
        for frame in range(10):         # natural order, sorted order, split order, replicas, then fp64 at 3 / 2 / 3 / 2 waves per SIMD, the kept one
            st = {}
            got, ids = sc.render(cam, precision=prec, want_ids=True, stats=st)
            np.testing.assert_array_equal(got, plain)
            np.testing.assert_array_equal(ids, ids0)
            assert st["rays"] == s0["rays"]        # per-wave count slots + sum_rays_kernel against the workgroups' atomics of the plain launch
            assert st["shadow_rays_culled"] == s0["shadow_rays_culled"]
    # a different view on the same handle starts over from whatever order is there: still identical
    cam2 = cam.resized(cam.hsize // 2, cam.vsize // 2)
    a, _ = sc.render(cam2, precision="fp64", feedback=False)
    b, _ = sc.render(cam2, precision="fp64")
    np.testing.assert_array_equal(a, b)


def test_jittered_area_light(scenes):
    """shadow_glamour_shot.yml as shipped (jitter: true).  The reference draws from Python's global RNG in a
    process-timing-dependent order, so there is no image to compare with; the kernel's counter-based samples must be
    reproducible for a seed, differ between seeds, leave every pixel whose light cells are all lit or all shadowed in
    the unjittered frame (nearly) alone, and average out to the unjittered penumbra."""
    cam, _, sj = scenes("shadow_glamour_shot")
    _, _, sn = scenes("shadow_glamour_shot_nojitter")
    cam = cam.resized(240, 96)
    plain, ids_plain = sn.render(cam, precision="fp64", want_ids=True)
    a, ids_a = sj.render(cam, precision="fp64", want_ids=True, seed=1)
    b, _ = sj.render(cam, precision="fp64", seed=1)
    c, _ = sj.render(cam, precision="fp64", seed=2)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ids_a, ids_plain)
    assert np.abs(a - c).max() > 1e-3
    mean = np.mean([sj.render(cam, precision="fp64", seed=s)[0] for s in range(3, 19)], axis=0)
    assert np.abs(a - plain).mean() < 0.01
    # averaging 16 seeds removes most of the noise; what stays is the bias of cell centres against uniform samples
    assert np.abs(mean - plain).mean() < 0.75 * np.abs(a - plain).mean()
    assert (np.abs(a - plain).max(axis=2) > 0.05).mean() < 0.2                # the penumbra is a minority of the frame


def test_tiles_to_canvas_on_device(scenes):
    """The multi-GPU frame assembly on one GPU: two "ranks" trace interleaved tile lists into compact buffers, the
    buffers sit back to back like a gathered one (with padding slots), rtc_scatter_tiles_device builds the canvas."""
    import torch
    from ray_tracer_challenge_amd import dist
    cam, world, sc = scenes("cover")
    cam = cam.resized(150, 70)                                   # ragged: 5 x 3 tiles of 32, the last ones partial
    st = {}
    want, want_ids = sc.render(cam, precision="fp64", want_ids=True, stats=st)
    dev = torch.device("cuda", 0)
    fgs = [dist.FrameGather(cam.hsize, cam.vsize, 2, r, dev, with_ids=True) for r in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    counters = [torch.zeros(13, dtype=torch.int64, device=dev) for _ in range(2)]
    for fg, rays in zip(fgs, counters):
        sc.render_device(cam, fg.compact.data_ptr(), ids_ptr=fg.compact_ids.data_ptr(), rays_ptr=rays.data_ptr(), tile_ids_ptr=fg.tiles_dev.data_ptr(),
                         n_tiles=len(fg.mine), tile_size=fg.tile, stream=stream)
    root = fgs[0]
    for r, fg in enumerate(fgs):
        root.all_tiles[r].copy_(fg.compact)                      # what the two RCCL gathers do
        root.all_ids[r].copy_(fg.compact_ids)
    root.scatter()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(root.frame.cpu().numpy(), want)
    np.testing.assert_array_equal(root.ids.cpu().numpy(), want_ids)
    total = (counters[0] + counters[1]).cpu().numpy()            # what dist.reduce_counters leaves on rank 0
    assert total[1:4].tolist() == st["rays"][1:] and total[12] == st["shadow_rays_culled"]
    assert total[0] >= cam.hsize * cam.vsize                     # primary rays: every pixel of every (ragged) tile


@pytest.mark.parametrize("name,size", [("cover", None), ("shadow_glamour_shot_nojitter", (121, 47))])
def test_ppm_encoded_on_the_device(name, size, scenes):
    """rtc_render_ppm: both of main.py's files (final_clamped.ppm, final.ppm) byte for byte what construct_ppm makes of
    the fp64 frame (canvas.py:133-184), including the brightness-scaled variant's frame maximum."""
    from ray_tracer_challenge_amd.canvas import construct_ppm
    cam, world, sc = scenes(name)
    if size:
        cam = cam.resized(*size)
    rgb, _ = sc.render(cam, precision="fp64")
    clamped, scaled = sc.render_ppm(cam, precision="fp64")
    assert clamped == construct_ppm(rgb)
    assert scaled == construct_ppm(rgb, False)
    if name == "cover":
        assert rgb.max() > 1.0 and clamped != scaled            # the scene the reference ships with components above 1
    only, none = sc.render_ppm(cam, precision="fp64", scaled=False)
    assert only == clamped and none is None


def test_ppm_files_against_the_reference(scenes):
    """north_star's gate is stated on the PPM output.  tests/golden/cover_ppm.npz holds the two files the REFERENCE writes
    for cover/cover.yml (render_multi -> construct_ppm(ca) = final_clamped.ppm, construct_ppm(ca, False) = final.ppm,
    main.py:334-354); rtc_render_ppm's two files of the same scene: same length and header, every byte within 1, at least
    99.9 % of the bytes equal -- in fp64 (measured: all equal) and in fp32 (the gate)."""
    g = load_golden("cover_ppm")
    cam, world, sc = scenes("cover")
    cam = cam.resized(g["meta"]["hsize"], g["meta"]["vsize"])
    for prec, need in (("fp64", 0.9999), ("fp32", 0.999)):
        clamped, scaled = sc.render_ppm(cam, precision=prec)
        for mine, ref in ((clamped, g["clamped"]), (scaled, g["scaled"])):
            a, b = np.frombuffer(bytes(mine), np.uint8).astype(int), ref.astype(int)
            assert a.shape == b.shape
            np.testing.assert_array_equal(a[:15], b[:15])           # "P6\n100 100\n255\n"
            assert np.abs(a - b).max() <= 1 and (a == b).mean() >= need, (prec, np.abs(a - b).max(), (a == b).mean())


def _same_image(a, b, ia, ib):
    """EXTRA check, not the parity gate (that is _against_oracle): two conservative hierarchies over the same triangles give
    identical images except where two leaves are hit at exactly the same t (shared edges), which either tree may report first."""
    assert (np.abs(a - b).max(axis=2) > 1e-9).mean() <= 2e-4
    assert (ia != ib).mean() <= 2e-4


def _oracle_sample(cache, oracle_cls, key, world, cam, stride):
    """The oracle's colours and hit leaves on the strided pixel sample of `cam`, computed once per session per key."""
    if key not in cache:
        px, py = _sample(cam, stride)
        orc = oracle_cls(world)
        want, leaf, _, _ = orc.render_pixels(cam, px, py)
        cache[key] = (orc, px, py, want, leaf)
    return cache[key]


def _against_oracle(label, sample, cam, rgb, ids, allowed=0.002):
    """The f-2 gate: a frame traced through a GPU-built or refitted tree is held to the ORACLE exactly as
    test_fp64_matches_oracle holds the host-built tree: every differing pixel classified, none unexplained, few explained."""
    orc, px, py, want, leaf = sample
    unexplained, classes = explain_mismatches(orc, cam, px, py, rgb[px, py], want, ids[px, py], leaf)
    record_classes(label, len(px), classes, len(unexplained))
    assert not unexplained, (label, unexplained[:20])
    assert sum(classes.values()) <= allowed * len(px) + 2, (label, classes)


@pytest.mark.parametrize("name,size,stride", [("teapot", None, 3), ("dragons", (480, 192), 4), ("cover", None, 1)])
def test_triangle_bvh_built_on_the_gpu(name, size, stride, oracle_cls):
    """rtc_scene_build_bvh_device: the frame traced through the GPU-built triangle tree against the ORACLE
    (same sample and same rule as test_fp64_matches_oracle: every differing pixel a tie / grazing / acne / edge pixel), in
    fp32 within the gate; extra: same image as with the host-built tree; scenes without triangles are left alone."""
    cam, world, _ = load_pack(name)
    if size:
        cam = cam.resized(*size)
    sample = _oracle_sample(_ORACLE_SAMPLES, oracle_cls, (name, cam.hsize, cam.vsize, stride), world, cam, stride)
    with _ffi.Scene(compile_scene(world, instancing=False), device=0) as sc:
        s0, s1 = {}, {}
        a, ia = sc.render(cam, precision="fp64", want_ids=True, stats=s0)
        a32, _ = sc.render(cam, precision="fp32")
        before = sc.info()
        sc.build_bvh_on_device()
        after = sc.info()
        b, ib = sc.render(cam, precision="fp64", want_ids=True, stats=s1)
        b32, ib32 = sc.render(cam, precision="fp32", want_ids=True)
        _against_oracle("gpu-built tree: %s %dx%d" % (name, cam.hsize, cam.vsize), sample, cam, b, ib)
        assert frac_off_by_more_than_one(b32[sample[1], sample[2]], sample[3]) <= 0.001
        _same_image(a, b, ia, ib)
        assert frac_off_by_more_than_one(b32, a) <= 0.001
        assert abs(sum(s0["rays"]) - sum(s1["rays"])) <= 1e-4 * sum(s0["rays"])
        if name == "cover":
            assert after == before                               # no triangles: nothing to rebuild
        else:
            assert after["bvh_depth"] <= 31 and after["bvh_nodes"] > 0
        sc.build_bvh_on_device()                                 # twice in a row: rebuilt from the same rows
        d, idd = sc.render(cam, precision="fp64", want_ids=True)
        np.testing.assert_array_equal(b, d)
        if sc.cs.counts["n_tris"]:
            sc.update_triangles(sc.cs.arrays["tri_geom"], sc.cs.arrays["tri_norm"], refit=True)      # the GPU-built tree refitted around the same rows
            e, ie = sc.render(cam, precision="fp64", want_ids=True)
            _against_oracle("gpu-built tree refitted in place: %s %dx%d" % (name, cam.hsize, cam.vsize), sample, cam, e, ie)


@pytest.mark.parametrize("name,size,stride", [("dragons", (480, 192), 4), ("teapot", None, 3)])
def test_scene_created_with_build_on_device(name, size, stride, oracle_cls):
    """rtc_scene_desc::build_flags = RTC_BUILD_ON_DEVICE (what render() / render_multi() create their one-frame scenes with):
    no host SAH build for meshes of 16 384 triangles and more, the tree comes from the GPU builder -- held to the ORACLE like
    every other tree, and a refit and a rebuild on such a handle work like on any other."""
    cam, world, _ = load_pack(name)
    if size:
        cam = cam.resized(*size)
    sample = _oracle_sample(_ORACLE_SAMPLES, oracle_cls, (name, cam.hsize, cam.vsize, stride), world, cam, stride)
    cs = compile_scene(world, instancing=False)
    with _ffi.Scene(cs, device=0, build_on_device=True) as sc:
        info = sc.info()
        assert info["bvh_nodes"] > 0 and info["bvh_depth"] <= 31
        if cs.counts["n_tris"] >= 16384:
            assert info["bvh_max_leaf"] == 1                      # the GPU builder's trees (the host's hold two triangles a leaf)
        a, ia = sc.render(cam, precision="fp64", want_ids=True)
        a32, _ = sc.render(cam, precision="fp32")
        _against_oracle("scene created with RTC_BUILD_ON_DEVICE: %s %dx%d" % (name, cam.hsize, cam.vsize), sample, cam, a, ia)
        assert frac_off_by_more_than_one(a32[sample[1], sample[2]], sample[3]) <= 0.001
        sc.update_triangles(cs.arrays["tri_geom"], cs.arrays["tri_norm"], refit=True)
        b, ib = sc.render(cam, precision="fp64", want_ids=True)
        _against_oracle("... refitted: %s" % name, sample, cam, b, ib)
        sc.build_bvh_on_device()
        c, ic = sc.render(cam, precision="fp64", want_ids=True)
        _against_oracle("... rebuilt: %s" % name, sample, cam, c, ic)


def test_instanced_dragons_against_the_oracle(oracle_cls):
    """Instanced meshes (rtc_scene_desc::inst; compile_scene(world, instancing=True)): bounding_boxes.yml loads dragon.obj six
    times, the mesh is kept once in object space with six instances.  The frame is held to the ORACLE like every other (fp64:
    every differing pixel classified, none unexplained; fp32 within the gate), hit ids keep the depth-first leaf numbering,
    the flattened form gives the same frame to rounding and the same ray counts, and the scene takes a third of the memory."""
    cam, world, _ = load_pack("dragons")
    cam = cam.resized(480, 192)
    sample = _oracle_sample(_ORACLE_SAMPLES, oracle_cls, ("dragons", cam.hsize, cam.vsize, 4), world, cam, 4)
    flat, inst = compile_scene(world, instancing=False), compile_scene(world, instancing=True)
    assert inst.counts["n_instances"] == 6 and inst.counts["n_meshes"] == 1 and inst.counts["n_mesh_tris"] == 23490 and inst.counts["n_tris"] == 0
    assert flat.counts["n_instances"] == 0 and flat.counts["n_tris"] == 140940 and flat.counts["n_leaves"] == inst.counts["n_leaves"]
    with _ffi.Scene(flat, device=0) as sc:
        s0 = {}
        a, ia = sc.render(cam, precision="fp64", want_ids=True, stats=s0)
        bytes_flat = sc.info()["device_bytes_fp64"]
    with _ffi.Scene(inst, device=0) as sc:
        s1 = {}
        b, ib = sc.render(cam, precision="fp64", want_ids=True, stats=s1)
        b32, ib32 = sc.render(cam, precision="fp32", want_ids=True)
        assert sc.last_launch()["csg"] == 3                      # the kernel variants that walk instances
        assert sc.info()["device_bytes_fp64"] < 0.4 * bytes_flat
        _against_oracle("instanced dragons %dx%d" % (cam.hsize, cam.vsize), sample, cam, b, ib)
        assert frac_off_by_more_than_one(b32[sample[1], sample[2]], sample[3]) <= 0.001
        _same_image(a, b, ia, ib)
        assert s0["rays"] == s1["rays"] and s0["shadow_rays_culled"] == s1["shadow_rays_culled"]
        for k in range(6):                                       # later frames (launch order, split parts, replicas): the same image
            c, ic = sc.render(cam, precision="fp64", want_ids=True)
            np.testing.assert_array_equal(b, c)
            np.testing.assert_array_equal(ib, ic)


def test_instanced_meshes_among_other_things(oracle_cls):
    """Two meshes, three and two instances, under shear / rotation / non-uniform scale; one instance of glass (the n1 / n2
    container pass walks instances too), one that casts no shadow, a patterned one (pattern space = the instance's object
    space); spheres, a plane and loose triangles beside them.  Full frame against the oracle, instanced and flattened."""
    import io
    from ray_tracer_challenge_amd import shapes as S
    from ray_tracer_challenge_amd.canvas import color
    from ray_tracer_challenge_amd.matrix import rotation_x, rotation_y, scaling, shearing, translation
    from ray_tracer_challenge_amd.obj_parser import OBJParser, obj_to_group
    from ray_tracer_challenge_amd.renderer import Camera, PointLight, World, view_transform
    from ray_tracer_challenge_amd.vector import point, vector

    def ball_obj(n_lat, n_lon, squash):
        """OBJ text of a tessellated ellipsoid with vertex normals: 2 n_lat n_lon triangles"""
        lines, idx = [], {}
        for i in range(n_lat + 1):
            for j in range(n_lon):
                th, ph = np.pi * i / n_lat, 2 * np.pi * j / n_lon
                x, y, z = np.sin(th) * np.cos(ph), np.cos(th) * squash, np.sin(th) * np.sin(ph)
                idx[(i, j)] = len(idx) + 1
                lines.append("v %.9f %.9f %.9f" % (x, y, z))
                lines.append("vn %.9f %.9f %.9f" % (x, y / squash ** 2, z))
        lines.append("g ball")
        for i in range(n_lat):
            for j in range(n_lon):
                a, b, c, d = idx[(i, j)], idx[(i, (j + 1) % n_lon)], idx[(i + 1, (j + 1) % n_lon)], idx[(i + 1, j)]
                lines.append("f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c))
                lines.append("f %d//%d %d//%d %d//%d" % (a, a, c, c, d, d))
        return "\n".join(lines)

    def mesh(text, material, transform):
        g = obj_to_group(OBJParser(io.StringIO(text), default_material=material))
        g.transform = transform
        return g
    big, small = ball_obj(24, 24, 0.6), ball_obj(23, 23, 1.3)          # 1152 and 1058 triangles
    glass = S.Material(color=color(0.1, 0.1, 0.2), diffuse=0.2, transparency=0.9, reflective=0.3, refractive_index=1.4)
    ghost = S.Material(color=color(0.9, 0.8, 0.1), casts_shadow=False)
    striped = S.Material(pattern=S.Stripe(color(1, 0, 0), color(1, 1, 1), scaling(0.25, 0.25, 0.25) * rotation_y(0.5)))
    plain = S.Material(color=color(0.2, 0.7, 0.3), reflective=0.2)
    things = [
        S.Plane(S.Material(color=color(0.6, 0.6, 0.6), reflective=0.1), translation(0, -1.5, 0)),
        mesh(big, plain, translation(-2.2, 0, 0.5) * shearing(0.3, 0, 0, 0, 0, 0.2)),
        mesh(big, glass, translation(0, 0.1, 0) * rotation_x(0.6) * scaling(1.1, 1.4, 0.9)),
        mesh(big, striped, translation(2.3, 0.2, 1.0) * rotation_y(1.1)),
        S.Group(S.Material(), translation(0, 0, 3), [mesh(small, ghost, translation(-1.2, 0.8, 0) * scaling(0.8, 0.8, 0.8)),
                                                     mesh(small, plain, translation(1.4, 0.6, 0.5) * scaling(0.7, 1.0, 0.7))]),
        S.Sphere(S.Material(color=color(0.8, 0.2, 0.2)), translation(0, 0.1, 0) * scaling(0.4, 0.4, 0.4)),      # inside the glass mesh
        S.Triangle(S.Material(color=color(0.3, 0.3, 0.8)), None, point(-3, -1.5, 4), point(3, -1.5, 4), point(0, 3, 4.5)),
    ]
    world = World([PointLight(point(-6, 8, -8), color(0.9, 0.9, 0.9)), PointLight(point(5, 3, -6), color(0.3, 0.3, 0.3))], things)
    cam = Camera(96, 64, 1.0, view_transform(point(0, 1.5, -7), point(0, 0.2, 0), vector(0, 1, 0)))
    inst, flat = compile_scene(world, instancing=True), compile_scene(world, instancing=False)
    assert inst.counts["n_instances"] == 5 and inst.counts["n_meshes"] == 2 and inst.counts["n_tris"] == 1
    px, py = np.meshgrid(np.arange(cam.hsize), np.arange(cam.vsize), indexing="ij")
    px, py = px.ravel().astype(np.int32), py.ravel().astype(np.int32)
    orc = oracle_cls(world)
    want, leaf, _, _ = orc.render_pixels(cam, px, py)
    frames = {}
    for label, cs in (("instanced", inst), ("flattened", flat)):
        with _ffi.Scene(cs, device=0) as sc:
            rgb, ids = sc.render(cam, precision="fp64", want_ids=True)
            rgb32, _ = sc.render(cam, precision="fp32")
            _against_oracle("two meshes, five instances (%s)" % label, (orc, px, py, want, leaf), cam, rgb, ids, allowed=0.004)
            assert frac_off_by_more_than_one(rgb32[px, py], want) <= 0.002
            frames[label] = (rgb, ids)
    assert np.mean(np.any(np.abs(frames["instanced"][0] - frames["flattened"][0]) > 1e-9, axis=2)) <= 0.004
    assert np.mean(frames["instanced"][1] != frames["flattened"][1]) <= 0.004


def _moved_teapot(fn):
    """The teapot pack with every triangle vertex moved by `fn` (object space): the world (for the oracle) and its
    compiled form (whose tri_geom / tri_norm rows are what rtc_scene_update_triangles takes)."""
    cam, world, _ = load_pack("teapot")
    move_triangles(world, fn)
    return world, compile_scene(world)


@pytest.mark.parametrize("refit", [False, True], ids=["rebuild", "refit"])
def test_moving_triangles_rebuild_on_the_gpu(refit, oracle_cls):
    """rtc_scene_update_triangles / rtc_scene_refit_triangles: the teapot moved by new triangle rows + GPU rebuild (or
    refit of the host-built tree), held to the ORACLE of the moved world on every pixel; extra: like a scene created
    from the moved world (host-built tree)."""
    cam, world, _ = load_pack("teapot")
    cam = cam.resized(160, 80)
    cs = compile_scene(world)
    off = np.array([2.5, -1.5, 1.0])                             # object units (the teapot spans ~30)
    moved_world, moved = _moved_teapot(lambda p: p + off)
    assert moved.arrays["tri_geom"].shape == cs.arrays["tri_geom"].shape
    sample = _oracle_sample({}, oracle_cls, "moved", moved_world, cam, 1)
    with _ffi.Scene(cs, device=0) as sc:
        still, _ = sc.render(cam, precision="fp64")
        sc.update_triangles(moved.arrays["tri_geom"], moved.arrays["tri_norm"], refit=refit)
        got, ids = sc.render(cam, precision="fp64", want_ids=True)
        got2, _ = sc.render(cam, precision="fp64")                # launch-order feedback restarted for the new tree
        got32, _ = sc.render(cam, precision="fp32")
    _against_oracle("teapot moved, %s" % ("refit of the host tree" if refit else "rebuilt on the GPU"), sample, cam, got, ids)
    assert frac_off_by_more_than_one(got32[sample[1], sample[2]], sample[3]) <= 0.001
    with _ffi.Scene(moved, device=0) as sc2:
        want, ids_want = sc2.render(cam, precision="fp64", want_ids=True)
    assert np.abs(want - still).max() > 0.05                      # it did move
    _same_image(want, got, ids_want, ids)
    np.testing.assert_array_equal(got, got2)


def test_refit_follows_a_deforming_mesh(oracle_cls):
    """rtc_scene_refit_triangles on both kinds of tree: every triangle vertex of the teapot displaced by its own amount (a
    deformation that tears the mesh, not a rigid motion), refitted twice in a row (host SAH tree, then the GPU-built tree),
    both precisions, held to the ORACLE of the deformed world on every pixel; extra: against a scene created from the
    deformed world."""
    cam, world, _ = load_pack("teapot")
    cam = cam.resized(200, 100)
    cs = compile_scene(world)

    def deformed(amount, seed):
        rng = np.random.default_rng(seed)
        return _moved_teapot(lambda p: p + amount * np.sin(0.4 * p[:, [1, 2, 0]]) + 0.02 * amount * rng.standard_normal(p.shape))

    (w1, c1), (w2, c2) = deformed(1.5, 5), deformed(4.0, 6)
    g1, n1, g2, n2 = c1.arrays["tri_geom"], c1.arrays["tri_norm"], c2.arrays["tri_geom"], c2.arrays["tri_norm"]
    s1 = _oracle_sample({}, oracle_cls, "d1", w1, cam, 1)
    s2 = _oracle_sample({}, oracle_cls, "d2", w2, cam, 1)

    def fresh(c):
        with _ffi.Scene(c, device=0) as sc2:
            return sc2.render(cam, precision="fp64", want_ids=True)
    want1, ids1 = fresh(c1)
    want2, ids2 = fresh(c2)
    assert np.abs(want1 - want2).max() > 0.05
    with _ffi.Scene(cs, device=0) as sc:
        before = sc.info()
        sc.update_triangles(g1, n1, refit=True)                                # host SAH tree refitted
        assert sc.info() == before                                             # same tree, new boxes
        a, ia = sc.render(cam, precision="fp64", want_ids=True)
        a32, _ = sc.render(cam, precision="fp32")
        _against_oracle("teapot deformed 1.5, host tree refitted", s1, cam, a, ia)
        _same_image(want1, a, ids1, ia)
        assert frac_off_by_more_than_one(a32[s1[1], s1[2]], s1[3]) <= 0.001
        sc.update_triangles(g2, n2, refit=True)                                # and again, further
        b, ib = sc.render(cam, precision="fp64", want_ids=True)
        _against_oracle("teapot deformed 4.0, host tree refitted twice", s2, cam, b, ib)
        _same_image(want2, b, ids2, ib)
        sc.build_bvh_on_device()                                               # GPU-built tree over g2 ...
        c, ic = sc.render(cam, precision="fp64", want_ids=True)
        _against_oracle("teapot deformed 4.0, rebuilt on the GPU", s2, cam, c, ic)
        sc.update_triangles(g1, n1, refit=True)                                # ... refitted back to g1
        d, idd = sc.render(cam, precision="fp64", want_ids=True)
        _against_oracle("teapot deformed 1.5, GPU-built tree refitted", s1, cam, d, idd)
        _same_image(want1, d, ids1, idd)
        # rows handed over in device memory: same tree and rows as the host-pointer calls give, so the same frames
        import torch
        tg = torch.from_numpy(np.ascontiguousarray(g2, dtype=np.float64)).to("cuda:0")
        tn = torch.from_numpy(np.ascontiguousarray(n2, dtype=np.float64)).to("cuda:0")
        torch.cuda.synchronize()
        sc.update_triangles_device(tg.data_ptr(), tn.data_ptr(), len(g2), refit=True)
        e, ie = sc.render(cam, precision="fp64", want_ids=True)
        e32, _ = sc.render(cam, precision="fp32")
        _against_oracle("teapot deformed 4.0, rows from device memory, refit", s2, cam, e, ie)
        sc.update_triangles(g2, n2, refit=True)
        f, _ = sc.render(cam, precision="fp64")
        f32, _ = sc.render(cam, precision="fp32")
        np.testing.assert_array_equal(e, f)
        np.testing.assert_array_equal(e32, f32)
        _same_image(want2, e, ids2, ie)
        sc.update_triangles_device(tg.data_ptr(), tn.data_ptr(), len(g2), refit=False)
        h, ih = sc.render(cam, precision="fp64", want_ids=True)
        _against_oracle("teapot deformed 4.0, rows from device memory, rebuild", s2, cam, h, ih)
        _same_image(want2, h, ids2, ih)


def test_result_arrays_come_from_the_pinned_pool(scenes):
    """_ffi draws result arrays from page-locked blocks (rtc_host_alloc): two results never share memory while both are
    alive, a released block is handed out again, and views keep their block alive."""
    import gc
    cam, world, sc = scenes("cover")
    cam = cam.resized(256, 128)                       # 786 KB frame: above the pool's small-array cut-off
    a, _ = sc.render(cam, precision="fp64")
    keep = a.copy()
    b, _ = sc.render(cam, precision="fp32")
    assert a.ctypes.data != b.ctypes.data
    np.testing.assert_array_equal(a, keep)            # the second render did not write into the first result
    view = a[10:20]
    addr = a.ctypes.data
    del a
    gc.collect()
    c, _ = sc.render(cam, precision="fp64")
    assert c.ctypes.data != addr                      # `view` still holds the block
    np.testing.assert_array_equal(view, keep[10:20])
    del view
    gc.collect()
    d, _ = sc.render(cam, precision="fp64")
    assert d.ctypes.data == addr                      # released block reused
    np.testing.assert_array_equal(d, keep)
    # the C entry points themselves
    p = C.c_void_p()
    assert _ffi.lib().rtc_host_alloc(1 << 20, C.byref(p)) == 0 and p.value
    _ffi.lib().rtc_host_free(p)


@pytest.mark.parametrize("tile", [8, 16, 64, 128])
def test_other_tile_sizes_render_the_same_frame(tile, scenes):
    """rtc_render_tiles with tile sizes other than the default 32 (any multiple of 8): every tile, then a strided subset,
    scattered into a canvas, equal the plain frame pixel for pixel -- in both launch orders (the second call of a view
    runs dearest-block-first)."""
    cam, world, sc = scenes("cover")
    cam = cam.resized(150, 90)                                           # not a multiple of any of the tile sizes
    want, want_ids = sc.render(cam, precision="fp64", want_ids=True)
    tx, ty = _ffi.tiles_for(cam.hsize, cam.vsize, tile)
    for subset in (np.arange(tx * ty, dtype=np.int32), np.arange(tx * ty, dtype=np.int32)[::3]):
        for _ in range(3):                                               # plain order, recorded order, steady state
            part, part_ids = sc.render_tiles(cam, subset, precision="fp64", tile_size=tile, want_ids=True)
            rgb, ids = np.full((cam.hsize, cam.vsize, 3), -5.0), np.full((cam.hsize, cam.vsize), -5, np.int32)
            _ffi.scatter_tiles(rgb, ids, subset, part, part_ids, tile)
            covered = ids != -5
            np.testing.assert_array_equal(rgb[covered], want[covered])
            np.testing.assert_array_equal(ids[covered], want_ids[covered])
            if len(subset) == tx * ty:
                assert covered.all()


def test_ppm_quantisation_on_synthetic_frames():
    """rtc_quantize_device / rtc_frame_max_device against canvas.quantize (reference canvas.py:133-156) on frames made of
    the values where the rule could go wrong: negatives, zeros, values above the maximum, exact k/255 and (k + 0.5)/255
    steps and their neighbours one ulp away -- clamped and scaled by the frame maximum, odd sizes."""
    import torch
    rng = np.random.default_rng(3)
    for w, h in ((1, 1), (7, 5), (33, 64), (130, 3)):
        steps = rng.integers(0, 256, (w, h, 3)) / 255.0
        halves = (rng.integers(0, 255, (w, h, 3)) + 0.5) / 255.0
        pick = rng.integers(0, 6, (w, h, 3))
        frame = np.where(pick == 0, steps, np.where(pick == 1, halves, np.where(pick == 2, np.nextafter(halves, 0.0),
                 np.where(pick == 3, np.nextafter(halves, 2.0), np.where(pick == 4, rng.uniform(-0.5, 0.0, (w, h, 3)), rng.uniform(0.0, 1.7, (w, h, 3)))))))
        frame = np.ascontiguousarray(frame, np.float64)
        dev = torch.from_numpy(frame).to("cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        top = torch.zeros(1, dtype=torch.float64, device="cuda:0")
        assert _ffi.lib().rtc_frame_max_device(C.c_void_p(dev.data_ptr()), frame.size, C.c_void_p(top.data_ptr()), C.c_void_p(stream)) == 0
        torch.cuda.synchronize()
        assert float(top[0]) == frame.max()
        for clamp in (True, False):
            out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
            maxc = 1.0 if clamp else float(frame.max())
            assert _ffi.lib().rtc_quantize_device(C.c_void_p(dev.data_ptr()), w, h, maxc, C.c_void_p(out.data_ptr()), C.c_void_p(stream)) == 0
            torch.cuda.synchronize()
            np.testing.assert_array_equal(out.cpu().numpy(), quantize(frame, clamp))


def test_one_scene_handle_rendered_from_two_threads(scenes):
    """a scene handle may be rendered from several host threads, one HIP stream per call.  Two threads render
    two different views of one handle at the same time -- through rtc_render_device on streams of their own and through
    rtc_render -- frame after frame (so that launch-order feedback is recorded, sorted and reused while the other thread
    does the same for its view); every frame must equal the one a single thread gets."""
    import threading
    import torch
    cam, world, sc = scenes("cover")
    views = [cam.resized(160, 100), cam.resized(96, 128)]
    want = [sc.render(v, precision="fp64", want_ids=True, feedback=False) for v in views]
    dev = torch.device("cuda", 0)
    errors = []

    def worker(k):
        try:
            v = views[k]
            stream = torch.cuda.Stream(device=dev)
            frame = torch.zeros((v.hsize, v.vsize, 3), dtype=torch.float64, device=dev)
            ids = torch.zeros((v.hsize, v.vsize), dtype=torch.int32, device=dev)
            for it in range(12):
                frame.fill_(-1.0)
                torch.cuda.synchronize()
                sc.render_device(v, frame.data_ptr(), ids_ptr=ids.data_ptr(), precision="fp64", stream=stream.cuda_stream)
                stream.synchronize()
                if not np.array_equal(frame.cpu().numpy(), want[k][0]) or not np.array_equal(ids.cpu().numpy(), want[k][1]):
                    errors.append("device path, thread %d frame %d" % (k, it))
                got, got_ids = sc.render(v, precision="fp64", want_ids=True)
                if not np.array_equal(got, want[k][0]) or not np.array_equal(got_ids, want[k][1]):
                    errors.append("host path, thread %d frame %d" % (k, it))
        except Exception as e:          # noqa: BLE001 - reported below
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_jittered_area_light_against_the_reference(scenes):
    """SURVEY 8d config 3.  shadow_glamour_shot.yml AS SHIPPED (jitter: true, 10 x 10 area light) was rendered by the
    reference's single-process path for 8 seeds of Python's RNG (oracle/make_golden.py make_jitter ->
    tests/golden/shadow_glamour_shot_jitter.npz: 8 x 8 block means per seed, their mean and their spread).  The reference's
    jittered frames are not reproducible pixel for pixel by anything but itself, so the check is statistical: the mean over
    16 kernel seeds of the kernel's block means is within 1/255 of the reference's mean block means everywhere (the
    reference's own seed-to-seed spread of a block mean is at most 0.0014, a third of that), and the kernel's noise has
    the reference's size."""
    g = load_golden("shadow_glamour_shot_jitter")
    cam, world, sc = scenes("shadow_glamour_shot")
    cam = cam.resized(g["meta"]["hsize"], g["meta"]["vsize"])
    b = g["meta"]["block"]

    def block_means(frame):
        w, h = frame.shape[0] // b * b, frame.shape[1] // b * b
        return frame[:w, :h].reshape(w // b, b, h // b, b, 3).mean(axis=(1, 3))

    assert g["block_std"].max() < 1 / 255.0                                  # else the bound below would have to be 3 sigma
    mine = np.array([block_means(sc.render(cam, precision="fp64", seed=s)[0]) for s in range(16)])
    assert np.abs(mine.mean(axis=0) - g["block_mean"]).max() < 1 / 255.0
    assert np.abs(mine.mean(axis=0) - g["block_mean"]).mean() < 0.25 / 255.0
    # per-seed noise of a block mean: same size as the reference's (both are 100 uniform samples per light cell grid)
    mine_std, ref_std = mine.std(axis=0, ddof=1), g["block_std"]
    assert 0.5 * ref_std.mean() < mine_std.mean() < 2.0 * ref_std.mean()
    # and single frames: every pixel of the reference's seed-0 frame within the penumbra noise of the kernel's frames
    lo, hi = np.min([sc.render(cam, precision="fp64", seed=s)[0] for s in range(16)], axis=0), None
    hi = np.max([sc.render(cam, precision="fp64", seed=s)[0] for s in range(16)], axis=0)
    inside = (g["frame_seed0"] >= lo - 0.08) & (g["frame_seed0"] <= hi + 0.08)
    assert inside.mean() > 0.999
    mine32 = np.array([block_means(sc.render(cam, precision="fp32", seed=s)[0]) for s in range(16)])
    assert np.abs(mine32.mean(axis=0) - g["block_mean"]).max() < 1 / 255.0
