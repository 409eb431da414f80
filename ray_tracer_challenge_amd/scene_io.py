"""Scene packs: a whole scene (camera, lights, config, shape tree, materials, patterns,
meshes, textures) in one `.npz`, loadable without the YAML / OBJ / image files it came from.

Used for shipping benchmark and test scenes to machines that have no scene sources, and as a
fast path past OBJ parsing.  The pack holds this package's own description of the scene (a
JSON tree + numeric arrays), so `load_scene(save_scene(...))` compiles to the same buffers.
Runs of triangles inside a group are stored as one `(n, 18)` array (p1 p2 p3 n1 n2 n3).
"""
import hashlib
import json

import numpy as np

from . import shapes as S
from .canvas import TextureCanvas
from .config import GlobalConfig
from .matrix import Matrix
from .renderer import AreaLight, Camera, PointLight, World
from .vector import Tuple4, _arr

_SIMPLE = {"Sphere": S.Sphere, "Plane": S.Plane, "Cube": S.Cube}
_CAPPED = {"Cylinder": S.Cylinder, "Cone": S.Cone}
_TWO_COLOR = {"Pattern": S.Pattern, "Checkers": S.Checkers, "Gradient": S.Gradient,
              "RadialGradient": S.RadialGradient, "Ring": S.Ring, "Stripe": S.Stripe}
_OPS = {"CSGUnion": S.CSGUnion, "CSGIntersect": S.CSGIntersect, "CSGDifference": S.CSGDifference}
_UVMAPS = {"PlaneUVMap": S.PlaneUVMap, "SphereUVMap": S.SphereUVMap,
           "CylinderUVMap": S.CylinderUVMap, "CubeUVMap": S.CubeUVMap}


def _name(x):
    return x.__name__ if isinstance(x, type) else type(x).__name__


def _m(x):
    return [float(v) for v in _arr(x).reshape(-1)]


class _Saver(object):
    def __init__(self):
        self.arrays = {}
        self.materials, self._mat = [], {}
        self.patterns, self._pat = [], {}
        self._tex = {}
        self._mesh_of = {}

    def material(self, m):
        if id(m) not in self._mat:
            self._mat[id(m)] = len(self.materials)
            self.materials.append(None)
            self.materials[self._mat[id(m)]] = {
                "color": _m(m.color), "ambient": float(m.ambient), "diffuse": float(m.diffuse),
                "specular": float(m.specular), "shininess": float(m.shininess),
                "reflective": float(m.reflective), "transparency": float(m.transparency),
                "refractive_index": float(m.refractive_index), "casts_shadow": bool(m.casts_shadow),
                "pattern": None if m.pattern is None else self.pattern(m.pattern)}
        return self._mat[id(m)]

    def pattern(self, p):
        if id(p) in self._pat:
            return self._pat[id(p)]
        idx = self._pat[id(p)] = len(self.patterns)
        self.patterns.append(None)
        name = type(p).__name__
        d = {"type": name, "transform": _m(p.transform), "a": _m(p.a), "b": _m(p.b)}
        if name == "BlendedPattern":
            d["sub"] = [self.pattern(p.p1), self.pattern(p.p2)]
        elif name == "NestedPattern":
            d["sub"] = [self.pattern(p.p1), self.pattern(p.p2), self.pattern(p.p3)]
        elif name == "PerturbedPattern":
            d["sub"] = [self.pattern(p.p1)]
            d["freq"], d["scale_factor"], d["octaves"] = float(p.freq), float(p.scale_factor), int(p.octaves)
        elif name == "TextureMapPattern":
            d["sub"] = [self.pattern(p.uv_pattern)]
            d["uv_map"] = _name(p.uv_map)
        elif name in ("CubeMapPattern", "CylinderMapPattern"):
            d["sub"] = [self.pattern(f) for f in p.faces]
        elif name == "UVChecker":
            d["width"], d["height"] = float(p.width), float(p.height)
        elif name == "UVAlignCheck":
            d["colors"] = [[float(v) for v in c] for c in (p.main, p.ul, p.ur, p.bl, p.br)]
        elif name == "UVTexturePattern":
            tc = p.canvas
            if id(tc) not in self._tex:
                self._tex[id(tc)] = "tex_%d" % len(self._tex)
                self.arrays[self._tex[id(tc)]] = np.asarray(tc.pixels)
            d["texture"] = self._tex[id(tc)]
        elif name not in _TWO_COLOR:
            raise ValueError("cannot serialise pattern " + name)
        self.patterns[idx] = d
        return idx

    def node(self, o):
        name = type(o).__name__
        d = {"kind": name, "transform": _m(o.transform), "material": self.material(o.material)}
        if name in _CAPPED:
            d["min"], d["max"], d["closed"] = float(o.minimum), float(o.maximum), bool(o.closed)
            for k in ("min", "max"):
                if np.isinf(d[k]):
                    d[k] = "inf" if d[k] > 0 else "-inf"
        elif name in ("Triangle", "SmoothTriangle"):
            d["tri"] = self._tri_rows([o]).tolist()
        elif name == "Group":
            kids, run = [], []
            for c in o.children:
                if self._packable(c, run):
                    run.append(c)
                    continue
                self._flush(run, kids)
                kids.append(self.node(c))
            self._flush(run, kids)
            d["children"] = kids
        elif name == "CSG":
            d["op"] = _name(o.op)
            d["left"], d["right"] = self.node(o.left), self.node(o.right)
        elif name not in _SIMPLE:
            raise ValueError("cannot serialise shape " + name)
        return d

    @staticmethod
    def _packable(c, run):
        if type(c).__name__ != "SmoothTriangle" or not np.array_equal(_arr(c.transform), np.eye(4)):
            return False
        return not run or c.material is run[0].material

    @staticmethod
    def _tri_rows(tris):
        out = np.zeros((len(tris), 18))
        for i, t in enumerate(tris):
            out[i, 0:3], out[i, 3:6], out[i, 6:9] = _arr(t.p1)[:3], _arr(t.p2)[:3], _arr(t.p3)[:3]
            if hasattr(t, "n1"):
                out[i, 9:12], out[i, 12:15], out[i, 15:18] = _arr(t.n1)[:3], _arr(t.n2)[:3], _arr(t.n3)[:3]
        return out

    def _flush(self, run, kids):
        if run:
            rows = self._tri_rows(run)
            digest = hashlib.sha1(rows.tobytes()).hexdigest()
            key = self._mesh_of.get(digest)
            if key is None:     # identical meshes (several instances of one OBJ) are stored once
                key = self._mesh_of[digest] = "mesh_%d" % len(self._mesh_of)
                self.arrays[key] = rows
            kids.append({"kind": "SmoothTriangleRun", "array": key, "material": self.material(run[0].material)})
            del run[:]


def save_scene(path, camera, world, config=None, meta=None):
    sv = _Saver()
    doc = {"format": "rtc-scene-pack-1", "meta": meta or {},
           "camera": {"hsize": int(camera.hsize), "vsize": int(camera.vsize),
                      "field_of_view": float(camera.field_of_view), "transform": _m(camera.transform)},
           "lights": [], "config": None, "contains": [sv.node(o) for o in world.contains]}
    for l in world.lights or []:
        if hasattr(l, "position"):
            doc["lights"].append({"type": "point", "position": _m(l.position), "intensity": _m(l.intensity)})
        else:
            doc["lights"].append({"type": "area", "corner": _m(l.corner), "uvec": _m(l.uvec), "vvec": _m(l.vvec),
                                  "usteps": int(l.usteps), "vsteps": int(l.vsteps), "jitter": bool(l.jitter),
                                  "intensity": _m(l.intensity)})
    if config is not None:
        doc["config"] = {"thread_count": config.thread_count, "timeout": config.timeout,
                         "divide_threshold": config.divide_threshold, "clamping": bool(config.clamping)}
    doc["materials"], doc["patterns"] = sv.materials, sv.patterns
    arrays = dict(sv.arrays)
    arrays["scene_json"] = np.frombuffer(json.dumps(doc).encode("utf-8"), dtype=np.uint8)
    np.savez_compressed(path, **arrays)


class _Loader(object):
    def __init__(self, doc, arrays):
        self.doc, self.arrays = doc, arrays
        self._mat, self._pat, self._tex = {}, {}, {}

    def material(self, i):
        if i not in self._mat:
            d = dict(self.doc["materials"][i])
            pat = d.pop("pattern")
            d["color"] = np.array(d["color"])
            self._mat[i] = S.Material(pattern=None if pat is None else self.pattern(pat), **d)
        return self._mat[i]

    def pattern(self, i):
        if i in self._pat:
            return self._pat[i]
        d = self.doc["patterns"][i]
        name, xf = d["type"], Matrix(np.array(d["transform"]).reshape(4, 4))
        a, b = np.array(d["a"]), np.array(d["b"])
        sub = [self.pattern(k) for k in d.get("sub", [])]
        if name in _TWO_COLOR:
            p = _TWO_COLOR[name](a, b, xf)
        elif name == "BlendedPattern":
            p = S.BlendedPattern(sub[0], sub[1])
        elif name == "NestedPattern":
            p = S.NestedPattern(sub[0], sub[1], sub[2])
        elif name == "PerturbedPattern":
            p = S.PerturbedPattern(sub[0], d["freq"], d["scale_factor"], d["octaves"])
        elif name == "TextureMapPattern":
            p = S.TextureMapPattern(sub[0], _UVMAPS[d["uv_map"]], xf)
        elif name == "CubeMapPattern":
            p = S.CubeMapPattern(uv_right=sub[0], uv_left=sub[1], uv_up=sub[2], uv_down=sub[3],
                                 uv_front=sub[4], uv_back=sub[5], transform=xf)
        elif name == "CylinderMapPattern":
            p = S.CylinderMapPattern(uv_body=sub[0], uv_cap=sub[1], transform=xf)
        elif name == "UVChecker":
            p = S.UVChecker(a, b, d["width"], d["height"])
        elif name == "UVAlignCheck":
            p = S.UVAlignCheck(*d["colors"])
        elif name == "UVTexturePattern":
            key = d["texture"]
            if key not in self._tex:
                self._tex[key] = TextureCanvas(pixels=self.arrays[key])
            p = S.UVTexturePattern(self._tex[key])
        else:
            raise ValueError("unknown pattern type in pack: " + name)
        self._pat[i] = p
        return p

    def node(self, d):
        name = d["kind"]
        if name == "SmoothTriangleRun":
            return self._tris(self.arrays[d["array"]], self.material(d["material"]))
        xf, mat = Matrix(np.array(d["transform"]).reshape(4, 4)), self.material(d["material"])
        if name in _SIMPLE:
            return [_SIMPLE[name](mat, xf)]
        if name in _CAPPED:
            return [_CAPPED[name](mat, xf, float(d["min"]), float(d["max"]), d["closed"])]
        if name in ("Triangle", "SmoothTriangle"):
            r = np.array(d["tri"])[0]
            pts = [Tuple4(np.append(r[k:k + 3], 1.0)) for k in (0, 3, 6)]
            if name == "Triangle":
                return [S.Triangle(mat, xf, *pts)]
            return [S.SmoothTriangle(mat, xf, *pts, *[Tuple4(np.append(r[k:k + 3], 0.0)) for k in (9, 12, 15)])]
        if name == "Group":
            g = S.Group(mat, xf, [])
            for c in d["children"]:
                if c["kind"] == "SmoothTriangleRun":       # the compiler flattens such a run from the array itself (shapes.TriangleRun)
                    rows = self.arrays[c["array"]]
                    g.add_triangle_run(self._tris(rows, self.material(c["material"])), rows[:, :18])
                else:
                    g.add_child(self.node(c))
            return [g]
        if name == "CSG":
            return [S.CSG(mat, xf, _OPS[d["op"]], self.node(d["left"])[0], self.node(d["right"])[0])]
        raise ValueError("unknown shape kind in pack: " + name)

    @staticmethod
    def _tris(rows, mat):
        ident = S.matrix4x4identity()
        p = np.ones((len(rows), 3, 4))
        p[:, :, :3] = rows[:, 0:9].reshape(-1, 3, 3)
        n = np.zeros((len(rows), 3, 4))
        n[:, :, :3] = rows[:, 9:18].reshape(-1, 3, 3)
        return [S.SmoothTriangle(mat, ident, Tuple4(p[i, 0]), Tuple4(p[i, 1]), Tuple4(p[i, 2]),
                                 Tuple4(n[i, 0]), Tuple4(n[i, 1]), Tuple4(n[i, 2])) for i in range(len(rows))]


def load_scene(path):
    """-> {'camera', 'lights', 'world': [top-level shapes], 'config', 'meta'} (the dict shape of
    `yaml_file_to_world_objects`, plus `meta`)."""
    with np.load(path, allow_pickle=False) as z:
        arrays = {k: z[k] for k in z.files}
    doc = json.loads(arrays.pop("scene_json").tobytes().decode("utf-8"))
    if doc.get("format") != "rtc-scene-pack-1":
        raise ValueError("not a scene pack: " + str(path))
    ld = _Loader(doc, arrays)
    c = doc["camera"]
    cam = Camera(c["hsize"], c["vsize"], c["field_of_view"], Matrix(np.array(c["transform"]).reshape(4, 4)))
    lights = []
    for l in doc["lights"]:
        if l["type"] == "point":
            lights.append(PointLight(Tuple4(l["position"]), np.array(l["intensity"])))
        else:
            al = AreaLight(Tuple4(l["corner"]), Tuple4(l["uvec"]) * l["usteps"], l["usteps"],
                           Tuple4(l["vvec"]) * l["vsteps"], l["vsteps"], l["jitter"], np.array(l["intensity"]))
            al.uvec, al.vvec = Tuple4(l["uvec"]), Tuple4(l["vvec"])  # exact per-step vectors
            lights.append(al)
    contains = []
    for d in doc["contains"]:
        contains.extend(ld.node(d))
    cfg = None if doc["config"] is None else GlobalConfig(**doc["config"])
    return {"camera": cam, "lights": lights, "world": contains, "config": cfg, "meta": doc.get("meta", {})}


def load_world(path):
    objs = load_scene(path)
    return objs["camera"], World(objs["lights"], objs["world"]), objs
