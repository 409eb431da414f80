"""Scene model: shapes, materials, patterns (drop-in names for the reference's `shapes.py`).

These classes are *descriptions only*.  Nothing here intersects a ray: the per-ray work
of the reference (`local_intersect`, `local_normal_at`, `pattern_at`, reference
shapes.py:525-878, 1614-2033) runs in the HIP kernels after `compile.compile_scene`
has flattened a tree of these objects (or of the reference's own objects, which expose
the same attributes) into device buffers.  What does run on the host, once per scene, is
the bounding-box bookkeeping behind `Group.divide` (reference shapes.py:10-135, 480-507).

Attribute names follow the reference so either object tree can be compiled:
`transform(.arr)`, `material`, `parent`, `children`, `left/right/op`, `minimum/maximum/
closed`, `p1..p3`, `n1..n3`, and on materials `color, ambient, diffuse, specular,
shininess, reflective, transparency, refractive_index, casts_shadow, pattern`.
"""
from copy import deepcopy

import numpy as np

from .canvas import color, TextureCanvas
from .matrix import Matrix, Transform, matrix4x4identity
from .vector import point, vector, _arr

EPSILON = 0.000001

BLACK = color(0, 0, 0)
WHITE = color(1, 1, 1)


# --------------------------------------------------------------------------------------
# bounding boxes (host only; used by Group.divide)
# --------------------------------------------------------------------------------------
class BoundingBox(object):
    def __init__(self, min=None, max=None):
        self.min = np.array([np.inf] * 3) if min is None else _arr(min)[:3].copy()
        self.max = np.array([-np.inf] * 3) if max is None else _arr(max)[:3].copy()

    def add_point(self, pt):
        p = _arr(pt)[:3]
        # comparisons (not fmin/fmax) so that NaN coordinates are ignored like the reference
        self.min = np.where(p < self.min, p, self.min)
        self.max = np.where(p > self.max, p, self.max)

    def add_box(self, other):
        if self is other:
            return
        self.add_point(other.min)
        self.add_point(other.max)

    def contains_point(self, pt):
        p = _arr(pt)[:3]
        return bool(np.all(self.min <= p) and np.all(p <= self.max))

    def contains_box(self, other):
        return self is other or (self.contains_point(other.min) and self.contains_point(other.max))

    def transform(self, m):
        m = _arr(m)
        lo, hi = self.min, self.max
        out = BoundingBox()
        for x in (lo[0], hi[0]):
            for y in (lo[1], hi[1]):
                for z in (lo[2], hi[2]):
                    with np.errstate(invalid="ignore"):
                        out.add_point(m @ np.array([x, y, z, 1.0]))
        return out

    def split_bounds(self):
        with np.errstate(invalid="ignore"):
            d = np.abs(self.max - self.min)
            greatest = max(d[0], d[1], d[2])
            lo = self.min.copy()
            hi = self.max.copy()
            axis = 0 if greatest == d[0] else (1 if greatest == d[1] else 2)
            lo[axis] = hi[axis] = lo[axis] + d[axis] / 2.0
        return BoundingBox(self.min, hi), BoundingBox(lo, self.max)

    def __repr__(self):
        return "BoundingBox: {} {}".format(self.min, self.max)


# --------------------------------------------------------------------------------------
# materials
# --------------------------------------------------------------------------------------
class Material(object):
    def __init__(self, color=None, ambient=0.1, diffuse=0.9, specular=0.9, shininess=200.0,
                 reflective=0.0, transparency=0.0, refractive_index=1.0, casts_shadow=True,
                 pattern=None):
        if ambient < 0 or diffuse < 0 or specular < 0 or shininess < 0:
            raise ValueError("Materials expect non-negative floating point values.")
        self.color = np.array([1.0, 1.0, 1.0]) if color is None else np.asarray(color, dtype=np.float64)
        self.ambient = np.float64(ambient)
        self.diffuse = np.float64(diffuse)
        self.specular = np.float64(specular)
        self.shininess = np.float64(shininess)
        self.pattern = pattern
        self.reflective = np.float64(reflective)
        self.transparency = np.float64(transparency)
        self.refractive_index = np.float64(refractive_index)
        self.casts_shadow = casts_shadow

    # YAML key -> constructor keyword.  `shadow` (not `casts-shadow`) is the key the
    # reference reads (shapes.py:190), so csg_test.yml's `casts-shadow` is ignored here too.
    _YAML_FLOATS = {"diffuse": "diffuse", "ambient": "ambient", "specular": "specular",
                    "shininess": "shininess", "reflective": "reflective",
                    "transparency": "transparency", "refractive-index": "refractive_index"}

    @classmethod
    def from_yaml(cls, obj):
        kw = {}
        if obj is not None:
            if "color" in obj:
                kw["color"] = color(*obj["color"])
            for key, name in cls._YAML_FLOATS.items():
                if key in obj:
                    kw[name] = float(obj[key])
            if "shadow" in obj:
                kw["casts_shadow"] = obj["shadow"]
            if "pattern" in obj:
                kw["pattern"] = Pattern.from_yaml(obj["pattern"])
        return cls(**kw)


# --------------------------------------------------------------------------------------
# shapes
# --------------------------------------------------------------------------------------
class Shape(object):
    def __init__(self, material=None, transform=None):
        self.transform = matrix4x4identity() if transform is None else transform
        self.material = Material() if material is None else material
        self.parent = None
        self._bounds = None

    @classmethod
    def _recursive_helper(cls, obj, defines):
        """YAML `add:` node -> shape (reference shapes.py:235-266)."""
        if "transform" not in obj:
            obj["transform"] = []
        kind = obj["add"]
        simple = {"sphere": Sphere, "plane": Plane, "cube": Cube, "cone": Cone,
                  "cylinder": Cylinder}
        if kind in simple:
            return simple[kind].from_yaml(obj)
        if kind == "group":
            if "material" in obj:
                for child in obj["children"]:
                    if "material" not in child:
                        child["material"] = deepcopy(obj["material"])
            return Group.from_yaml(obj, defines)
        if kind == "csg":
            if "material" in obj:
                for side in ("left", "right"):
                    if "material" not in obj[side]:
                        obj[side]["material"] = deepcopy(obj["material"])
            return CSG.from_yaml(obj, defines)
        if kind == "obj":
            from .obj_parser import OBJParser
            return OBJParser.from_yaml(obj, defines)
        return None

    @classmethod
    def from_yaml(cls, obj):
        return cls(material=Material.from_yaml(obj.get("material")),
                   transform=Transform.from_yaml(obj["transform"]))

    def includes(self, other):
        return self is other

    def _local_bounds(self):
        return BoundingBox([-1, -1, -1], [1, 1, 1])

    def bounds(self):
        # cached on first use, as the reference's lru_cache does (shapes.py:298-300)
        if self._bounds is None:
            self._bounds = self._local_bounds()
        return self._bounds

    def parent_space_bounds(self):
        return self.bounds().transform(self.transform)

    def divide(self, threshold):
        return


class Sphere(Shape):
    pass


class Plane(Shape):
    def _local_bounds(self):
        return BoundingBox([-np.inf, 0, -np.inf], [np.inf, 0, np.inf])


class Cube(Shape):
    pass


class _Capped(Shape):
    def __init__(self, material=None, transform=None, minimum=-np.inf, maximum=np.inf, closed=False):
        Shape.__init__(self, material, transform)
        self.minimum = float(minimum)
        self.maximum = float(maximum)
        self.closed = closed


class Cone(_Capped):
    @classmethod
    def from_yaml(cls, obj):
        return cls(material=Material.from_yaml(obj.get("material")),
                   transform=Transform.from_yaml(obj["transform"]),
                   minimum=obj["min"], maximum=obj["max"], closed=obj["closed"])

    def _local_bounds(self):
        limit = max(abs(self.minimum), abs(self.maximum))
        return BoundingBox([-limit, self.minimum, -limit], [limit, self.maximum, limit])


class Cylinder(_Capped):
    @classmethod
    def from_yaml(cls, obj):
        # a YAML cylinder is closed unless it says otherwise (reference shapes.py:725-727)
        return cls(material=Material.from_yaml(obj.get("material")),
                   transform=Transform.from_yaml(obj["transform"]),
                   minimum=obj["min"], maximum=obj["max"], closed=obj.get("closed", True))

    def _local_bounds(self):
        return BoundingBox([-1, self.minimum, -1], [1, self.maximum, 1])


class Triangle(Shape):
    def __init__(self, material, transform, p1, p2, p3):
        Shape.__init__(self, material, transform)
        self.p1, self.p2, self.p3 = p1, p2, p3

    @property
    def e1(self):
        return self.p2 - self.p1

    @property
    def e2(self):
        return self.p3 - self.p1

    def _local_bounds(self):
        b = BoundingBox()
        for p in (self.p1, self.p2, self.p3):
            b.add_point(p)
        return b


class SmoothTriangle(Triangle):
    def __init__(self, material, transform, p1, p2, p3, n1, n2, n3):
        Triangle.__init__(self, material, transform, p1, p2, p3)
        self.n1, self.n2, self.n3 = n1, n2, n3


class TriangleRun(object):
    """`count` consecutive children of a Group, from index `start`, that a loader made from one block of vertex data:
    rows[i] = p1, p2, p3, n1, n2, n3 (18 numbers) of child start + i -- SmoothTriangles with the identity transform and
    one shared material.  The scene compiler flattens such a run from `rows` in one NumPy step instead of reading six
    attributes of each of (for the dragons) 140 940 triangle objects; it first checks that the run still is what the
    loader made -- the same objects at both ends and in the middle, with the same vertices -- and reads the objects
    otherwise.  Code that moves the triangles of a loaded mesh clears `group.triangle_runs`."""

    def __init__(self, start, triangles, rows):
        self.start, self.count, self.rows = int(start), len(triangles), rows
        self.probes = [(k, triangles[k]) for k in sorted({0, len(triangles) // 2, len(triangles) - 1})] if triangles else []
        self.material = triangles[0].material if triangles else None

    def still_valid(self, children):
        if self.count == 0 or self.start + self.count > len(children):
            return False
        for k, tri in self.probes:
            c = children[self.start + k]
            if c is not tri or c.material is not self.material or type(c).__name__ != "SmoothTriangle":
                return False
            got = np.concatenate([np.asarray(getattr(v, "arr", v), dtype=np.float64)[:3] for v in (c.p1, c.p2, c.p3, c.n1, c.n2, c.n3)])
            if not np.array_equal(got, self.rows[k]) or not np.array_equal(np.asarray(c.transform.arr), np.eye(4)):
                return False
        return True


class Group(Shape):
    """Children are kept in insertion order (the reference uses a set, so its order is
    arbitrary); order only matters for hits at exactly equal `t`."""

    def __init__(self, material=None, transform=None, children=()):
        Shape.__init__(self, material, transform)
        self.children = []
        self._child_ids = set()
        self.triangle_runs = []          # TriangleRun records of loader-made meshes among the children (may be empty)
        self.add_child(list(children))

    def add_triangle_run(self, triangles, rows):
        """Appends loader-made triangles (see TriangleRun) as children and remembers the block of vertex data behind them."""
        start = len(self.children)
        self.add_child(triangles)
        if len(self.children) == start + len(triangles):
            self.triangle_runs.append(TriangleRun(start, triangles, np.ascontiguousarray(rows, dtype=np.float64)))

    @classmethod
    def from_yaml(cls, tree, defines):
        mat = Material.from_yaml(tree["material"]) if "material" in tree else Material()
        xform = Transform.from_yaml(tree["transform"]) if "transform" in tree else matrix4x4identity()
        children = [Shape._recursive_helper(obj, defines) for obj in tree["children"] if "add" in obj]
        return cls(material=mat, transform=xform, children=children)

    def includes(self, other):
        return any(c.includes(other) for c in self.children)

    def add_child(self, shs):
        if self is shs:
            raise ValueError("Don't add a group to its own children collection.")
        if isinstance(shs, (set, list, tuple)):
            for sh in shs:
                self.add_child(sh)
        elif id(shs) not in self._child_ids:
            self._child_ids.add(id(shs))
            self.children.append(shs)
            shs.parent = self

    def _local_bounds(self):
        box = BoundingBox()
        for c in self.children:
            box.add_box(c.parent_space_bounds())
        return box

    # --- Group.divide, reference shapes.py:480-507, quirks included ------------------
    def divide(self, threshold):
        if threshold < len(self.children):
            left, right = self._partition_children()
            if len(left) > 0:
                self._make_subgroup(left)
            if len(right) > 0:
                self._make_subgroup(right)
        for c in list(self.children):
            c.divide(threshold)

    def _make_subgroup(self, children):
        # the sub-group is built with a copy of THIS group's transform (shapes.py:492):
        # harmless for identity, a double application otherwise.
        ids = set(id(c) for c in children)
        self.triangle_runs = []               # positions change
        self.children = [c for c in self.children if id(c) not in ids]
        self._child_ids -= ids
        self.add_child(Group(self.material, self.transform, children))

    def _partition_children(self):
        left_box, right_box = self.bounds().split_bounds()
        left, right = [], []
        for c in self.children:
            sub_box = c.bounds()  # the child's LOCAL bounds, not parent-space (shapes.py:502)
            if left_box.contains_box(sub_box):
                left.append(c)
            elif right_box.contains_box(sub_box):
                right.append(c)
        return left, right


class CSGUnion(object):
    name = "union"

    @classmethod
    def intersection_allowed(cls, lhit, inl, inr):
        return (lhit and not inr) or (not lhit and not inl)


class CSGIntersect(object):
    name = "intersection"

    @classmethod
    def intersection_allowed(cls, lhit, inl, inr):
        return (lhit and inr) or (not lhit and inl)


class CSGDifference(object):
    name = "difference"

    @classmethod
    def intersection_allowed(cls, lhit, inl, inr):
        return (lhit and not inr) or (not lhit and inl)


class CSG(Shape):
    def __init__(self, material, transform, op, left, right):
        Shape.__init__(self, material, transform)
        self.op = op
        self.left = left
        self.right = right
        self.left.parent = self
        self.right.parent = self

    @classmethod
    def from_yaml(cls, tree, defines):
        mat = Material.from_yaml(tree["material"]) if "material" in tree else Material()
        xform = Transform.from_yaml(tree["transform"]) if "transform" in tree else matrix4x4identity()
        sides = []
        for side in ("left", "right"):
            if "add" not in tree[side]:
                raise ValueError("Malformed CSG yaml", tree)
            sides.append(Shape._recursive_helper(tree[side], defines))
        op_name = tree["op"] if "op" in tree else tree["operation"]
        ops = {"difference": CSGDifference, "intersection": CSGIntersect, "union": CSGUnion}
        if op_name not in ops:
            raise ValueError("Malformed CSG yaml", tree)
        return cls(material=mat, transform=xform, op=ops[op_name], left=sides[0], right=sides[1])

    def includes(self, other):
        return self.left.includes(other) or self.right.includes(other)

    def _local_bounds(self):
        box = BoundingBox()
        for c in (self.left, self.right):
            box.add_box(c.parent_space_bounds())
        return box

    def divide(self, threshold):
        self.left.divide(threshold)
        self.right.divide(threshold)


# --------------------------------------------------------------------------------------
# patterns (descriptions; evaluated on the device)
# --------------------------------------------------------------------------------------
class Pattern(object):
    """Base class doubles as the book's test pattern: colour = pattern-space point."""

    def __init__(self, color1=WHITE, color2=BLACK, transform=None):
        self.transform = matrix4x4identity() if transform is None else transform
        self.a = color1
        self.b = color2

    @classmethod
    def from_yaml(cls, obj):
        xform = Transform.from_yaml(obj["transform"]) if "transform" in obj else matrix4x4identity()
        typ = obj["type"]
        two_color = {"checkers": Checkers, "gradient": Gradient, "radial-gradient": RadialGradient,
                     "rings": Ring, "stripe": Stripe}
        if typ in two_color:
            return two_color[typ](color1=color(*obj["colors"][0]), color2=color(*obj["colors"][1]),
                                  transform=xform)
        if typ == "blended":
            return BlendedPattern(Pattern.from_yaml(obj["left"]), Pattern.from_yaml(obj["right"]))
        if typ == "nested":
            return NestedPattern(Pattern.from_yaml(obj["primary"]), Pattern.from_yaml(obj["left"]),
                                 Pattern.from_yaml(obj["right"]))
        if typ == "perturbed":
            return PerturbedPattern(Pattern.from_yaml(obj["primary"]),
                                    frequency=np.float64(obj.get("frequency", 0.4)),
                                    scale_factor=np.float64(obj.get("scale-factor", 0.3)),
                                    octaves=obj.get("octaves", 1))
        if typ == "map":
            mapping = obj["mapping"]
            uv = Pattern.uv_from_yaml
            if mapping == "cube":
                # The reference's YAML path omits the `transform` argument here and raises
                # TypeError (shapes.py:1563-1569 vs 1843); identity is the only value it
                # could have meant, so that is what is passed (documented divergence).
                return CubeMapPattern(uv_left=uv(obj["left"]), uv_front=uv(obj["front"]),
                                      uv_right=uv(obj["right"]), uv_back=uv(obj["back"]),
                                      uv_up=uv(obj["up"]), uv_down=uv(obj["down"]),
                                      uv_map=CubeUVMap, transform=matrix4x4identity())
            if mapping == "cylindrical":
                if "uv_pattern" in obj:
                    cap, body = uv(obj["uv_pattern"]), uv(obj["uv_pattern"])
                else:
                    cap, body = uv(obj["top"]), uv(obj["front"])
                return CylinderMapPattern(uv_cap=cap, uv_body=body, uv_map=CylinderUVMap, transform=xform)
            if mapping == "planar":
                return TextureMapPattern(uv(obj["uv_pattern"]), PlaneUVMap, xform)
            if mapping == "spherical":
                return TextureMapPattern(uv(obj["uv_pattern"]), SphereUVMap, xform)
        raise ValueError("Unable to parse pattern type: {}".format(typ))

    @classmethod
    def uv_from_yaml(cls, obj):
        typ = obj["type"]
        if typ == "checkers":
            return UVChecker(color1=color(*obj["colors"][0]), color2=color(*obj["colors"][1]),
                             width=obj["width"], height=obj["height"])
        if typ == "align_check":
            c = obj["colors"]
            return UVAlignCheck(main_color=c["main"], upper_left=c["ul"], upper_right=c["ur"],
                                bottom_left=c["bl"], bottom_right=c["br"])
        if typ == "image":
            return UVTexturePattern(tcanvas=TextureCanvas(obj["file"]))
        raise ValueError("Unable to parse uv pattern type: {}".format(typ))


class Checkers(Pattern):
    pass


class Gradient(Pattern):
    pass


class RadialGradient(Pattern):
    pass


class Ring(Pattern):
    pass


class Stripe(Pattern):
    pass


class BlendedPattern(Pattern):
    def __init__(self, pattern1, pattern2=None):
        Pattern.__init__(self, pattern1.a, pattern1.b)
        self.p1 = pattern1
        self.p2 = pattern1 if pattern2 is None else pattern2


class NestedPattern(Pattern):
    def __init__(self, pattern1, pattern2=None, pattern3=None):
        Pattern.__init__(self, pattern1.a, pattern1.b)
        self.p1 = pattern1
        self.p2 = pattern1 if pattern2 is None else pattern2
        self.p3 = pattern1 if pattern3 is None else pattern3


class PerturbedPattern(Pattern):
    def __init__(self, pattern1, frequency=0.4, scale_factor=0.3, octaves=1):
        Pattern.__init__(self, pattern1.a, pattern1.b)
        self.p1 = pattern1
        self.freq = frequency
        self.scale_factor = scale_factor
        self.octaves = octaves


class SphereUVMap(object):
    pass


class CylinderUVMap(object):
    pass


class PlaneUVMap(object):
    pass


class CubeUVMap(object):
    pass


class TextureMapPattern(Pattern):
    def __init__(self, uv_pattern, uv_map, transform=None):
        Pattern.__init__(self, uv_pattern.a, uv_pattern.b, transform)
        self.uv_pattern = uv_pattern
        self.uv_map = uv_map


class CubeMapPattern(Pattern):
    def __init__(self, uv_left, uv_front, uv_right, uv_back, uv_up, uv_down, uv_map=CubeUVMap,
                 transform=None):
        Pattern.__init__(self, WHITE, BLACK, transform)
        self.faces = [uv_right, uv_left, uv_up, uv_down, uv_front, uv_back]  # order matters
        self.uv_map = uv_map


class CylinderMapPattern(Pattern):
    def __init__(self, uv_cap, uv_body, uv_map=CylinderUVMap, transform=None):
        Pattern.__init__(self, WHITE, BLACK, transform)
        self.faces = [uv_body, uv_cap]  # order matters
        self.uv_map = uv_map


class UVChecker(Pattern):
    def __init__(self, color1=WHITE, color2=BLACK, width=2, height=2):
        Pattern.__init__(self, color1, color2)
        self.width = width
        self.height = height


class UVAlignCheck(Pattern):
    def __init__(self, main_color, upper_left, upper_right, bottom_left, bottom_right):
        Pattern.__init__(self, WHITE, BLACK)
        self.main = main_color
        self.ul = upper_left
        self.ur = upper_right
        self.bl = bottom_left
        self.br = bottom_right


class UVTexturePattern(Pattern):
    def __init__(self, tcanvas):
        Pattern.__init__(self, WHITE, BLACK)
        self.canvas = tcanvas


# --------------------------------------------------------------------------------------
# convenience constructors with the reference's names
# --------------------------------------------------------------------------------------
def material():
    return Material(color(1, 1, 1), 0.1, 0.9, 0.9, 200.0)


def sphere():
    return Sphere(material(), matrix4x4identity())


def glass_sphere():
    s = sphere()
    s.material.transparency = np.float64(1.0)
    s.material.refractive_index = np.float64(1.5)
    return s


def plane():
    return Plane(Material(), matrix4x4identity())


def cube():
    return Cube(Material(), matrix4x4identity())


def cylinder():
    return Cylinder(material(), matrix4x4identity())


def cone():
    return Cone(material(), matrix4x4identity())


def group():
    return Group(Material(), matrix4x4identity(), [])


def triangle(p1, p2, p3):
    return Triangle(Material(), matrix4x4identity(), p1, p2, p3)


def test_pattern():
    return Pattern()


def stripe_pattern(a, b):
    return Stripe(a, b)
