"""Host logic: YAML define/extend macros, OBJ parsing, Group.divide, scene packs, the compiler's
output, PPM quantisation.  Where /root/reference is present (build container only) the loaders
are also cross-checked against the reference's own on its shipped scenes."""
import io
import os

import numpy as np
import pytest

from conftest import ROOT, load_pack
from ray_tracer_challenge_amd import shapes as S
from ray_tracer_challenge_amd.canvas import Canvas, color, construct_ppm, quantize
from ray_tracer_challenge_amd.compile import compile_scene, K_CSG_ROOT, K_PLANE
from ray_tracer_challenge_amd.matrix import Transform, rotation_x, scaling, translation
from ray_tracer_challenge_amd.obj_parser import OBJParser, obj_to_group
from ray_tracer_challenge_amd.renderer import World
from ray_tracer_challenge_amd.scene_io import load_world, save_scene
from ray_tracer_challenge_amd.vector import point
from ray_tracer_challenge_amd.yaml_parser import yaml_text_to_world_objects

YAML = """
- add: camera
  width: 40
  height: 20
  field-of-view: 0.7
  from: [0, 1, -5]
  to: [0, 0, 0]
  up: [0, 1, 0]
- add: light
  at: [-10, 10, -10]
  intensity: [1, 1, 1]
- add: config
  thread-count: 2
  divide-threshold: 3
- define: base-material
  value:
    color: [1, 0, 0]
    diffuse: 0.5
- define: shiny
  extend: base-material
  value:
    reflective: 0.25
- define: std
  value:
  - [ translate, 1, 0, 0 ]
  - [ scale, 2, 2, 2 ]
- define: big
  value:
  - std
  - [ scale, 3, 3, 3 ]
- define: leg
  value:
    add: group
    transform:
    - [ rotate-y, 0.5 ]
    children:
    - add: sphere
    - add: cylinder
      min: 0
      max: 1
- add: sphere
  material: shiny
  transform:
  - big
  - [ translate, 0, 1, 0 ]
- add: cube
  material:
    color: [0, 1, 0]
    shadow: false
- add: leg
  material:
    color: [0, 0, 1]
  transform:
  - [ translate, 0, 0, 2 ]
- add: csg
  op: difference
  left:
    add: cube
  right:
    add: sphere
    transform:
    - [ translate, 0.5, 0.5, 0 ]
"""


def test_yaml_define_extend_and_instancing():
    objs = yaml_text_to_world_objects(YAML)
    assert objs["camera"].hsize == 40 and objs["config"].thread_count == 2 and objs["config"].divide_threshold == 3
    root = objs["world"][0]
    assert type(root).__name__ == "Group" and len(root.children) == 4
    sph, cube, leg, csg = root.children
    # extend: parent copied, child keys overlaid (yaml_parser.py:36-45)
    assert np.allclose(sph.material.color, [1, 0, 0]) and sph.material.diffuse == 0.5 and sph.material.reflective == 0.25
    # nested transform defines are spliced in order; list applies first-to-last (matrix.py:128-132)
    want = translation(0, 1, 0).arr @ scaling(3, 3, 3).arr @ scaling(2, 2, 2).arr @ translation(1, 0, 0).arr
    assert np.allclose(sph.transform.arr, want)
    assert cube.material.casts_shadow is False
    # add: <define>: the define's transform entries go first; group material flows to children (shapes.py:249-254)
    assert np.allclose(leg.transform.arr, translation(0, 0, 2).arr @ Transform.from_yaml([["rotate-y", 0.5]]).arr)
    assert all(np.allclose(c.material.color, [0, 0, 1]) for c in leg.children)
    assert leg.children[1].closed is True           # YAML cylinders default to closed (shapes.py:725-727)
    assert csg.op is S.CSGDifference and csg.left.parent is csg


def test_obj_parser_fan_and_groups():
    text = """
v -1 1 0
v -1.0000 0.0000 0.0000
v 1 0 0
v 1 1 0
v 0 2 0
vn 0 0 1
vn 0 0 1
vn 0 0 1

g First
f 1//1 2//2 3//3 4//1 5//2
g Second
f 1/9/1 3/9/2 4/9/3
g Empty
"""
    p = OBJParser(io.StringIO(text))
    assert len(p.vertices) == 6 and len(p.normals) == 4
    g = obj_to_group(p)
    names = [len(c.children) for c in g.children]
    assert names == [3, 1]                           # fan of a pentagon = 3 triangles; empty groups dropped
    t = g.children[0].children[1]
    assert type(t).__name__ == "SmoothTriangle"
    assert t.p1.compare(point(-1, 1, 0)) and t.p2.compare(point(1, 0, 0)) and t.p3.compare(point(1, 1, 0))
    with pytest.raises(ValueError):
        OBJParser(io.StringIO("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"))   # the reference fails on normal-less faces too


def test_divide_partitions_like_the_reference():
    # three spheres left, one right, one straddling the split plane (shapes.py:480-507)
    g = S.group()
    xs = [-4, -3.5, -3, 4, 0]
    kids = []
    for x in xs:
        s = S.sphere()
        s.transform = translation(x, 0, 0)
        kids.append(s)
    g.add_child(kids)
    g.divide(1)
    # children's LOCAL bounds (unit cubes at the origin) straddle every split: nothing moves (shapes.py:502)
    assert len(g.children) == 5
    tris = [S.triangle(point(x, 0, 0), point(x + 0.5, 0, 0), point(x, 0.5, 0)) for x in (-4, -3, 3, 4)]
    g = S.group()
    g.add_child(tris)
    g.divide(1)
    assert [type(c).__name__ for c in g.children] == ["Group", "Group"]
    assert sorted(len(c.children) for c in g.children) == [2, 2]     # each half was split again into 1 + 1 sub-groups
    leaves = compile_scene(World([], [g])).counts["n_leaves"]
    assert leaves == 4


def test_divide_copies_parent_transform_quirk():
    g = S.Group(S.Material(), translation(0, 5, 0), [])
    tris = [S.triangle(point(x, 0, 0), point(x + 0.5, 0, 0), point(x, 0.5, 0)) for x in (-4, -3, 3, 4)]
    g.add_child(tris)
    before = compile_scene(World([], [g])).arrays["tri_geom"][:, 1].copy()
    g.divide(1)
    after = compile_scene(World([], [g])).arrays["tri_geom"][:, 1]
    # sub-groups carry a copy of the parent's transform (shapes.py:492): y moves by another +5 per level
    assert np.allclose(before, 5) and np.all(after >= 10)


def test_scene_pack_roundtrip(tmp_path):
    objs = yaml_text_to_world_objects(YAML)
    w = World(objs["lights"], objs["world"])
    a = compile_scene(w)
    path = str(tmp_path / "scene.npz")
    save_scene(path, objs["camera"], w, objs["config"], {"k": 1})
    cam, w2, o2 = load_world(path)
    b = compile_scene(w2)
    assert o2["meta"] == {"k": 1} and o2["config"].divide_threshold == 3 and cam.hsize == 40
    assert a.counts == b.counts
    for k in a.arrays:
        np.testing.assert_array_equal(a.arrays[k], b.arrays[k], err_msg=k)


def test_compiled_scene_structure():
    objs = yaml_text_to_world_objects(YAML)
    cs = compile_scene(World(objs["lights"], objs["world"]))
    A, C = cs.arrays, cs.counts
    assert C["n_leaves"] == 6 and C["n_csg_roots"] == 1 and C["n_csg_nodes"] == 1 and C["n_unbounded"] == 0
    assert list(A["csg_root"][0]) == [0, 1, 4, 6] and list(A["csg_node"][0][:4]) == [2, 4, 5, 6]
    kinds = (A["item_prim"].view(np.uint32) >> 28).tolist()
    assert kinds.count(K_CSG_ROOT) == 1 and len(kinds) == 5
    noshadow = ((A["item_prim"].view(np.uint32) >> 27) & 1).tolist()
    assert sum(noshadow) == 1                                        # the `shadow: false` cube
    assert np.all(A["item_lo"] <= A["item_hi"])
    cam, world, _ = load_pack("cover")
    cs = compile_scene(world)
    assert cs.counts["n_unbounded"] == 1 and (cs.arrays["unbounded_prim"].view(np.uint32)[0] >> 28) == K_PLANE


def test_triangles_move_to_world_space():
    t = S.triangle(point(0, 1, 0), point(-1, 0, 0), point(1, 0, 0))
    g = S.Group(S.Material(), scaling(2, 2, 2), [t])
    outer = S.Group(S.Material(), translation(0, 0, 5), [g])
    cs = compile_scene(World([], [outer]))
    geom = cs.arrays["tri_geom"][0]
    assert np.allclose(geom[0:3], [0, 2, 5]) and np.allclose(geom[3:6], [-2, -2, 0]) and np.allclose(geom[6:9], [2, -2, 0])
    assert np.isclose(geom[9], 1e-6 * 8)                             # EPSILON * |det(object->world)|
    n = cs.arrays["tri_norm"][0][:3]
    assert np.allclose(n / np.linalg.norm(n), [0, 0, -1])


def test_pattern_programs():
    W, B = color(1, 1, 1), color(0, 0, 0)
    s = S.sphere()
    s.material.pattern = S.NestedPattern(S.Checkers(W, B), S.Stripe(W, B), S.BlendedPattern(S.Ring(W, B), S.Gradient(W, B)))
    cs = compile_scene(World([], [s]))
    ops = cs.arrays["instr"][:cs.counts["n_instr"], 0].tolist()
    assert ops == [0, 0, 0, 2, 1]                                    # stripe, ring, gradient, BLEND, EVAL_AB(checkers)
    # PERTURB / RESTORE bracket the wrapped pattern's program (shapes.py:1924-1925); the node carries frequency, scale, octaves
    s.material.pattern = S.PerturbedPattern(S.BlendedPattern(S.Stripe(W, B), S.PerturbedPattern(S.Ring(W, B), 0.2, 0.1, 3)))
    cs = compile_scene(World([], [s]))
    ops = cs.arrays["instr"][:cs.counts["n_instr"], 0].tolist()
    assert ops == [3, 0, 3, 0, 4, 2, 4]
    outer, inner = cs.arrays["instr"][0, 1], cs.arrays["instr"][2, 1]
    assert cs.arrays["pat_type"][outer] == 8 and cs.arrays["pat_param"][outer][:3].tolist() == [0.4, 0.3, 1.0]
    assert cs.arrays["pat_param"][inner][:3].tolist() == [0.2, 0.1, 3.0]
    s.material.pattern = S.PerturbedPattern(S.Stripe(W, B), octaves=0)          # noise.pnoise3 raises for octaves < 1
    with pytest.raises(ValueError):
        compile_scene(World([], [s]))


def test_quantize_rule():
    ca = Canvas(2, 1)
    ca[0, 0] = color(1.5, 0.5, -0.2)
    ca[1, 0] = color(0.999, 0.001, 1.0)
    assert quantize(ca, True).tolist() == [[[255, 128, 0], [255, 0, 255]]]
    assert quantize(ca, False)[0, 0].tolist() == [255, 85, 0]       # scaled by the canvas maximum 1.5
    assert construct_ppm(ca).startswith(b"P6\n2 1\n255\n")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference not present (GPU box)")
@pytest.mark.parametrize("rel,cwd", [("cover/cover.yml", ""), ("csg_test.yml", ""), ("group_test/group.yml", ""),
                                     ("shadow_glamour_shot/shadow_glamour_shot.yml", ""),
                                     ("obj_file_test/obj_file_test.yml", "objects")])
def test_loader_matches_reference_loader(rel, cwd):
    """Same YAML through the reference's loader and ours -> identical compiled scenes (up to leaf order)."""
    from oracle import ref_harness as rh
    from oracle.make_golden import leaf_keys
    from ray_tracer_challenge_amd.yaml_parser import yaml_file_to_world_objects
    path = os.path.join(rh.REF, rel)
    with rh.chdir(os.path.join(rh.REF, cwd)):
        mine = yaml_file_to_world_objects(path)
    mods, objs, w = rh.load_reference_scene(path, cwd=os.path.join(rh.REF, cwd))
    a = compile_scene(World(mine["lights"], mine["world"]))
    b = compile_scene(w)
    assert a.counts == b.counts
    assert sorted(leaf_keys(a)) == sorted(leaf_keys(b))
    np.testing.assert_array_equal(a.arrays["light_geom"], b.arrays["light_geom"])
    ma = sorted(map(tuple, a.arrays["mat"][a.arrays["leaf_material"][:a.counts["n_leaves"]]].tolist()))
    mb = sorted(map(tuple, b.arrays["mat"][b.arrays["leaf_material"][:b.counts["n_leaves"]]].tolist()))
    assert ma == mb


MACRO_YAML = """
- add: camera
  width: 40
  height: 30
  field-of-view: 0.9
  from: [ 0, 2, -8 ]
  to: [ 0, 0.5, 0 ]
  up: [ 0, 1, 0 ]
- add: light
  at: [ -5, 8, -6 ]
  intensity: [ 1, 1, 1 ]
# dict defines, one extending the other, a third extending the second
- define: base-material
  value:
    color: [ 1, 1, 1 ]
    diffuse: 0.7
    ambient: 0.1
    specular: 0.2
- define: red-material
  extend: base-material
  value:
    color: [ 0.9, 0.2, 0.2 ]
    reflective: 0.25
- define: shiny-red-material
  extend: red-material
  value:
    specular: 0.9
    shininess: 300
# list defines: a define used inside a define, and a list that EXTENDS a list (the parent's items go in front, reversed)
- define: lift
  value:
  - [ translate, 0, 1, 0 ]
  - [ rotate-y, 0.4 ]
- define: lifted-and-shrunk
  value:
  - lift
  - [ scale, 0.5, 0.75, 0.5 ]
- define: skewed
  extend: lift
  value:
  - [ scale, 1.5, 0.5, 1 ]
  - [ rotate-z, 0.3 ]
# shape defines: a group with children that use defines, a csg whose operands are defines
- define: ball
  value:
    add: sphere
    material: red-material
    transform:
    - [ scale, 0.6, 0.6, 0.6 ]
- define: block
  value:
    add: cube
    material:
      color: [ 0.2, 0.3, 0.9 ]
      diffuse: 0.8
    transform:
    - [ scale, 0.5, 0.5, 0.5 ]
- define: pair
  value:
    add: group
    children:
    - add: ball
      transform:
      - [ translate, -1, 0, 0 ]
    - add: block
      transform:
      - lift
      - [ translate, 1, 0, 0 ]
    - add: cylinder
      min: 0
      max: 1
      closed: true
      material: shiny-red-material
      transform:
      - lifted-and-shrunk
- define: bitten
  value:
    add: csg
    op: difference
    left:
      add: block
    right:
      add: ball
      transform:
      - [ translate, 0.4, 0.4, -0.4 ]
# instances: plain, with transforms of their own (the define's come first), with a material of their own (it wins)
- add: plane
  material: base-material
  transform:
  - [ translate, 0, -1, 0 ]
- add: pair
- add: pair
  transform:
  - skewed
  - [ translate, 0, 0, 3 ]
- add: ball
  material:
    color: [ 0.1, 0.9, 0.1 ]
  transform:
  - lifted-and-shrunk
  - [ translate, 2.5, 0, 0 ]
- add: bitten
  transform:
  - [ rotate-y, 0.7 ]
  - [ translate, -2.5, 0.2, 0 ]
- add: cone
  min: -1
  max: 0
  closed: true
  material:
    red-material: ignored-by-both-loaders-except-as-an-overlay-key
    color: [ 0.5, 0.5, 0.1 ]
  transform:
  - skewed
  - [ translate, 0, 1.5, -2 ]
"""


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference not present (GPU box)")
def test_macro_language_matches_reference_loader(tmp_path):
    """A synthetic scene that exercises every macro rule -- dict `extend` (chained), list `extend` (the reference pushes the
    parent's items onto the front one by one, so they arrive REVERSED, yaml_parser.py:105-106), a define used inside a
    define, a group define whose children use defines, a csg define whose operands are defines, instance transforms and
    materials -- through the reference's loader and through ours: the macro-expanded YAML trees are equal entry for entry
    and the compiled scenes are identical."""
    import copy
    import yaml
    from oracle import ref_harness as rh
    from oracle.make_golden import leaf_keys
    from ray_tracer_challenge_amd import yaml_parser as mine
    path = tmp_path / "macros.yml"
    path.write_text(MACRO_YAML)
    mods, objs, w = rh.load_reference_scene(str(path), cwd=rh.REF)
    ours = mine.yaml_file_to_world_objects(str(path))
    a, b = compile_scene(World(ours["lights"], ours["world"])), compile_scene(w)
    assert a.counts == b.counts and a.counts["n_leaves"] == 11 and a.counts["n_csg_roots"] == 1
    # same leaves -- kind, collapsed world->object matrix to 1e-9, bounds -- with the same materials (the reference keeps
    # group children in hash-ordered sets, so the ORDER of leaves is its own)
    def leaves_with_materials(cs):
        mats = cs.arrays["mat"][cs.arrays["leaf_material"][:cs.counts["n_leaves"]]]
        return sorted(k + tuple(np.round(m, 12)) for k, m in zip(leaf_keys(cs), mats.tolist()))
    assert leaves_with_materials(a) == leaves_with_materials(b)
    np.testing.assert_array_equal(a.arrays["light_geom"], b.arrays["light_geom"])
    assert sorted(a.arrays["csg_node"][:, 0].tolist()) == sorted(b.arrays["csg_node"][:, 0].tolist())
    # the rewrite pass alone, on the raw trees: the reference's expand_defines_in_tree against _Macros.rewrite
    tree_a, tree_b = yaml.safe_load(MACRO_YAML), yaml.safe_load(MACRO_YAML)
    m = mine._Macros(tree_a)
    table_for_reference = copy.deepcopy(m.table)
    # (the table aliases the define entries' list values, in both implementations: rebuild that aliasing for tree_b)
    for entry in tree_b:
        if "define" in entry and isinstance(entry.get("value"), list):
            table_for_reference[entry["define"]] = entry["value"]
    m.rewrite(tree_a)
    mods["yaml_parser"].expand_defines_in_tree(tree_b, table_for_reference)
    assert tree_a == tree_b
    skewed = [e for e in tree_a if e.get("define") == "skewed"][0]["value"]
    assert skewed[:2] == [["rotate-y", 0.4], ["translate", 0, 1, 0]]          # the parent's items, reversed, in front


def _drop_runs(world):
    todo = list(world.contains)
    while todo:
        o = todo.pop()
        if hasattr(o, "children"):
            o.triangle_runs = []
            todo.extend(o.children)


def _same_compiled(a, b):
    assert a.counts == b.counts
    for k in a.arrays:
        assert a.arrays[k].dtype == b.arrays[k].dtype and np.array_equal(a.arrays[k], b.arrays[k]), k
    assert all(x is y for x, y in zip(a.leaf_objects, b.leaf_objects))


def test_triangle_runs_flatten_like_the_objects():
    """Loader-made meshes carry their vertex data as one block beside the triangle objects (shapes.TriangleRun) and
    compile_scene flattens them from the block: the arrays must be exactly what reading the objects one by one gives --
    for the scene packs, for OBJ text through the parser (several named groups, a transform above them, a primitive in
    between), and after Group.divide (which drops the runs) -- and a run whose triangles were moved is not trusted."""
    for name in ("teapot", "dragons"):
        cam, world, _ = load_pack(name)
        todo, in_runs = list(world.contains), 0
        while todo:
            o = todo.pop()
            if hasattr(o, "children"):
                in_runs += sum(r.count for r in o.triangle_runs)
                todo.extend(o.children)
        assert in_runs == {"teapot": 6320, "dragons": 140940}[name]
        fast = compile_scene(world)
        _drop_runs(world)
        _same_compiled(fast, compile_scene(world))
    rng = np.random.default_rng(4)
    lines = ["v %.6f %.6f %.6f" % tuple(v) for v in rng.uniform(-1, 1, (40, 3))] + ["vn %.6f %.6f %.6f" % tuple(v) for v in rng.uniform(-1, 1, (40, 3))]
    for gname in ("A", "B"):
        lines.append("g " + gname)
        for _ in range(30):
            k = rng.choice(np.arange(1, 41), size=int(rng.integers(3, 6)), replace=False)
            lines.append("f " + " ".join("%d//%d" % (i, i) for i in k))
    mesh = obj_to_group(OBJParser(io.StringIO("\n".join(lines)), default_material=S.Material(color=color(0.2, 0.5, 0.7))))
    mesh.transform = translation(0.5, -1, 2) * rotation_x(0.4) * scaling(1.5, 0.7, 1.1)
    assert all(len(g.triangle_runs) == 1 and g.triangle_runs[0].count == len(g.children) for g in mesh.children)
    world = World([], [S.Group(S.Material(), translation(0, 1, 0), [S.sphere(), mesh, S.Cube(S.Material(), scaling(2, 2, 2))])])
    fast = compile_scene(world)
    assert fast.counts["n_tris"] == sum(len(g.children) for g in mesh.children) > 100
    tri = mesh.children[1].children[len(mesh.children[1].children) // 2]
    tri.p2 = point(9, 9, 9)                              # a probe triangle moved: that group's run is not trusted any more
    moved = compile_scene(world)
    assert not np.array_equal(moved.arrays["tri_geom"], fast.arrays["tri_geom"])
    _drop_runs(world)
    _same_compiled(moved, compile_scene(world))
    mesh.divide(4)                                        # sub-groups are new groups: no runs, same leaves by another route
    assert all(not g.triangle_runs for g in mesh.children)
    assert compile_scene(world).counts["n_tris"] == fast.counts["n_tris"]


def test_repeated_meshes_become_instances():
    """compile_scene(world, instancing=True): the dragons pack loads one OBJ six times -- one object-space mesh, six instances
    whose leaves keep their depth-first numbers; nothing else about the scene changes; "auto" (the default) keeps the dragons
    flattened (too few triangles saved to pay for the slower fp64 walk); scenes without repeats are left alone."""
    cam, world, _ = load_pack("dragons")
    flat, inst, auto = compile_scene(world, instancing=False), compile_scene(world, instancing=True), compile_scene(world)
    assert auto.counts == flat.counts
    C, A = inst.counts, inst.arrays
    assert (C["n_instances"], C["n_meshes"], C["n_mesh_tris"], C["n_tris"]) == (6, 1, 23490, 0)
    assert C["n_leaves"] == flat.counts["n_leaves"] and C["n_items"] == flat.counts["n_items"] - 140940 + 6
    for name in ("leaf_kind", "leaf_xform", "leaf_material", "leaf_minmax", "leaf_closed", "xform", "mat", "light_geom"):
        np.testing.assert_array_equal(A[name], flat.arrays[name])
    kinds = (A["item_prim"].view(np.uint32) >> 28)
    assert (kinds == 9).sum() == 6 and (kinds == 5).sum() == 0
    for q, (mesh, xrow, leaf0, mat) in enumerate(A["inst"]):
        assert mesh == 0
        np.testing.assert_array_equal(A["leaf_tri"][leaf0:leaf0 + 23490], -2 - np.arange(23490))
        assert (A["leaf_xform"][leaf0:leaf0 + 23490] == xrow).all() and (A["leaf_material"][leaf0:leaf0 + 23490] == mat).all()
        # the instance's world box = the box of the flattened copy's triangles
        item = np.nonzero(A["item_prim"].view(np.uint32) == np.uint32((9 << 28) | q))[0][0]
        rows = flat.arrays["tri_geom"][flat.arrays["leaf_tri"][leaf0:leaf0 + 23490]]
        pts = np.concatenate([rows[:, 0:3], rows[:, 0:3] + rows[:, 3:6], rows[:, 0:3] + rows[:, 6:9]])
        np.testing.assert_allclose(A["item_lo"][item], pts.min(axis=0), atol=1e-12)
        np.testing.assert_allclose(A["item_hi"][item], pts.max(axis=0), atol=1e-12)
    # object-space rows through an instance's transform = the flattened copy's world-space rows
    leaf0, xrow = int(A["inst"][2][2]), int(A["inst"][2][1])
    m = np.vstack([A["xform"][xrow].reshape(3, 4), [0, 0, 0, 1]])
    a = np.linalg.inv(m)
    g = A["mesh_geom"]
    world_rows = flat.arrays["tri_geom"][flat.arrays["leaf_tri"][leaf0:leaf0 + 23490]]
    np.testing.assert_allclose(g[:, 0:3] @ a[:3, :3].T + a[:3, 3], world_rows[:, 0:3], atol=1e-12)
    np.testing.assert_allclose(g[:, 3:6] @ a[:3, :3].T, world_rows[:, 3:6], atol=1e-12)
    assert (g[:, 9] == 0.000001).all()
    for name in ("teapot", "cover"):
        cam, world, _ = load_pack(name)
        assert compile_scene(world, instancing=True).counts["n_instances"] == 0
