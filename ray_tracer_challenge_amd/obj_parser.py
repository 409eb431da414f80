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
        self.named_groups[self.current_group_name].add_child(self._fan_triangulation(tok))

    def _group(self, tok):
        name = tok[1]
        self.named_groups.setdefault(name, Group(self.default_material, matrix4x4identity(), []))
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
        out = []
        for (v2, n2), (v3, n3) in zip(corners[1:-1], corners[2:]):
            if n1 is None or n2 is None or n3 is None:
                raise ValueError("face without vertex normals: the reference always builds "
                                 "SmoothTriangle and fails on such a face too: " + " ".join(tok))
            out.append(SmoothTriangle(self.default_material, matrix4x4identity(),
                                      self.vertices[v1], self.vertices[v2], self.vertices[v3],
                                      self.normals[n1], self.normals[n2], self.normals[n3]))
        return out

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
    return outer


def obj_file_to_group(file_path, material=None):
    with open(file_path, "r") as f:
        return obj_to_group(OBJParser(f, default_material=material))
