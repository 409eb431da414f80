"""Wavefront OBJ -> Group of (smooth) triangles (drop-in for the reference's `obj_parser.py`).

Behaviour kept from the reference (obj_parser.py:18-128): `v` / `vn` / `f` / `g` records
only; 1-based indices; faces are fan-triangulated around their first vertex; a line is
split on single spaces and only the FIRST empty token is dropped (so `v  x y z` parses
and `v   x y z` does not); every face becomes a `SmoothTriangle` (the reference's
`len(self.normals) > 0` test is always true because index 0 holds a placeholder), which
means a face without `v/t/n` normal indices is an error; named groups with no faces are
dropped; the parsed groups are wrapped in one outer Group that carries the YAML transform.
"""
from .matrix import Transform, matrix4x4identity
from .shapes import Group, Material, SmoothTriangle
from .vector import point, vector


class OBJParser(object):
    DEFAULT = "##default_group"

    def __init__(self, open_file, default_material=None):
        self.default_material = Material() if default_material is None else default_material
        self.vertices = [None]
        self.normals = [None]
        self.named_groups = {self.DEFAULT: Group(self.default_material, matrix4x4identity(), [])}
        self.face_indices = {self.DEFAULT: []}          # per group: (v1, v2, v3, n1, n2, n3) of every triangle, in order
        self.current_group_name = self.DEFAULT
        self.default_group = self.named_groups[self.DEFAULT]
        for line in open_file:
            line = line.strip()
            if line:
                self._parse_line(line)

    def _parse_line(self, line):
        tok = line.split(" ")
        if "" in tok:
            tok.remove("")                               # the first empty token only (see the module docstring)
        record = self._RECORDS.get(tok[0])
        if record is not None:                           # every other record type is skipped
            record(self, tok)

    def _vertex(self, tok):
        self.vertices.append(point(*[float(c) for c in tok[1:4]]))

    def _normal(self, tok):
        self.normals.append(vector(*[float(c) for c in tok[1:4]]))

    def _face(self, tok):
        group = self.named_groups[self.current_group_name]
        before = len(group.children)
        tris, idx = self._fan_triangulation(tok)
        group.add_child(tris)
        if len(group.children) == before + len(tris):
            self.face_indices[self.current_group_name].extend(idx)

    def _group(self, tok):
        name = tok[1]
        self.named_groups.setdefault(name, Group(self.default_material, matrix4x4identity(), []))
        self.face_indices.setdefault(name, [])
        self.current_group_name = name

    _RECORDS = {"v": _vertex, "vn": _normal, "f": _face, "g": _group}

    def _corner(self, token):
        """'v/t/n' -> (vertex index, normal index or None)."""
        if "/" in token:
            parts = token.split("/")
            return int(parts[0]), int(parts[2])
        return int(token), None

    def _fan_triangulation(self, tok):
        corners = [self._corner(t) for t in tok[1:]]
        v1, n1 = corners[0]
        out, idx = [], []
        for (v2, n2), (v3, n3) in zip(corners[1:-1], corners[2:]):
            if n1 is None or n2 is None or n3 is None:
                raise ValueError("face without vertex normals: the reference always builds "
                                 "SmoothTriangle and fails on such a face too: " + " ".join(tok))
            out.append(SmoothTriangle(self.default_material, matrix4x4identity(),
                                      self.vertices[v1], self.vertices[v2], self.vertices[v3],
                                      self.normals[n1], self.normals[n2], self.normals[n3]))
            idx.append((v1, v2, v3, n1, n2, n3))
        return out, idx

    @classmethod
    def from_yaml(cls, obj, defines):
        mat = Material.from_yaml(obj["material"]) if "material" in obj else Material()
        xform = Transform.from_yaml(obj["transform"]) if "transform" in obj else matrix4x4identity()
        g = obj_file_to_group(obj["file"], mat)
        g.transform = xform
        return g


def obj_to_group(parser):
    outer = Group(Material(), matrix4x4identity(), [])
    outer.add_child([g for g in parser.named_groups.values() if len(g.children) > 0])
    _record_triangle_runs(parser)
    return outer


def _record_triangle_runs(parser):
    """Every named group's triangles as one block of vertex data beside the objects (shapes.TriangleRun): the scene
    compiler flattens a mesh from the block instead of from its tens of thousands of triangle objects."""
    import numpy as np
    from .shapes import TriangleRun
    if len(parser.vertices) < 2 or len(parser.normals) < 2:
        return
    V = np.array([np.asarray(p.arr, dtype=np.float64)[:3] for p in parser.vertices[1:]])
    N = np.array([np.asarray(n.arr, dtype=np.float64)[:3] for n in parser.normals[1:]])
    for name, g in parser.named_groups.items():
        idx = np.array(parser.face_indices.get(name, []), dtype=np.int64).reshape(-1, 6)
        if len(idx) == 0 or len(idx) != len(g.children) or g.triangle_runs:
            continue
        if (idx < 0).any():                               # negative (relative) indices resolve like Python's, as in the lists
            idx = np.where(idx < 0, idx + np.array([len(parser.vertices)] * 3 + [len(parser.normals)] * 3), idx)
        rows = np.concatenate([V[idx[:, 0] - 1], V[idx[:, 1] - 1], V[idx[:, 2] - 1], N[idx[:, 3] - 1], N[idx[:, 4] - 1], N[idx[:, 5] - 1]], axis=1)
        g.triangle_runs.append(TriangleRun(0, g.children, rows))


def obj_file_to_group(file_path, material=None):
    with open(file_path, "r") as f:
        return obj_to_group(OBJParser(f, default_material=material))
