"""YAML scene files -> camera, lights, world (drop-in for the reference's `yaml_parser.py`).

`yaml_file_to_world_objects(path)` returns the dict the reference returns (yaml_parser.py:10-64):
`{'camera', 'lights', 'world': [root Group], 'config'}`.

The scene language has macros: `- define: NAME` entries carry a `value` (a material dict, a list of
transform operations, or a whole shape description) and optionally `extend: PARENT`.  What the
reference does with them (yaml_parser.py:26-47 and 70-134) is a sequence of in-place rewrites whose
ORDER is observable, so this module states them as a small rewrite engine, `_Macros`:

  table      every define's value by name, in file order; a define that extends a dict-valued parent
             starts as the parent's entries overlaid with its own
  rules      applied to every top-level entry, for every define name in table order, one after the other:
               names in lists   the first occurrence of NAME in a `value` or `transform` list is replaced by the
                                items of NAME's value (the list object is edited in place: the table aliases the
                                define entries' lists, so later users see the expanded list)
               material         a string naming NAME (Python's `in`: a substring match on strings) becomes a
                                copy of NAME's value; a dict with a key NAME is laid over a copy of it
               extend           a dict value is laid over a copy of the parent's; a list value gets the parent's
                                items pushed onto its front ONE BY ONE, i.e. in reverse order
               instancing       `add: NAME` takes every key of a copy of NAME's value -- except that a material
                                the instance already has wins and the define's transform operations go in front
                                of the instance's own; the copy's group children / csg operands are expanded first
               children         a group entry's children are expanded with the same engine (once per define name,
                                as in the reference; the rewrites are idempotent)

Everything after the macros -- Camera / Light / GlobalConfig / Shape construction -- is the factories' business.
"""
from copy import deepcopy

import yaml

from .config import GlobalConfig
from .matrix import matrix4x4identity
from .renderer import Camera, Light
from .shapes import Group, Material, Shape

try:
    from yaml import CSafeLoader as _Loader
except ImportError:  # pragma: no cover
    from yaml import SafeLoader as _Loader


class _Macros(object):
    """The define table of one scene file and the rewrite rules that use it."""

    def __init__(self, entries):
        self.table = {}
        inherits = []
        for entry in entries:
            if "define" in entry:
                self.table[entry["define"]] = entry.get("value")
                if entry.get("extend") is not None:
                    inherits.append((entry["define"], entry["extend"]))
        for name, parent in inherits:                        # dict-valued parents only; list parents are the extend rule's
            if isinstance(self.table[parent], dict):
                self.table[name] = {**deepcopy(self.table[parent]), **self.table[name]}
        self.rules = (self._names_in("value"), self._material, self._names_in("transform"), self._extend, self._instance,
                      self._children)

    def copy_of(self, name):
        return deepcopy(self.table[name])

    def rewrite(self, entries):
        for entry in entries:
            for name in self.table:
                for rule in self.rules:
                    rule(entry, name)

    # ---- rules: (entry, define name) -> None, editing `entry` in place -------------------------------------------
    def _names_in(self, field):
        def rule(entry, name):
            seq = entry.get(field)
            if seq is not None and name in seq:
                at = seq.index(name)
                seq[at:at + 1] = self.copy_of(name)
        return rule

    def _material(self, entry, name):
        have = entry.get("material")
        if have is None or name not in have:
            return
        if isinstance(have, str):
            entry["material"] = self.copy_of(name)
        elif isinstance(have, dict):
            entry["material"] = {**self.copy_of(name), **have}

    def _extend(self, entry, name):
        if "extend" not in entry or name not in entry["extend"]:
            return
        del entry["extend"]
        if "value" not in entry:
            entry["value"] = self.copy_of(name)
            return
        own = entry["value"]
        if name in own:
            own[name] = self.copy_of(name)
        elif isinstance(own, dict):
            entry["value"] = {**self.copy_of(name), **own}
        else:
            own[0:0] = reversed(self.copy_of(name))          # pushed onto the front one by one (yaml_parser.py:105-106)

    def _instance(self, entry, name):
        if entry.get("add", None) != name or "add" not in entry:
            return
        proto = self.copy_of(name)
        kind = proto.get("add")
        if kind == "group" and "children" in proto:
            self.rewrite(proto["children"])
        if kind == "csg":
            for side in ("left", "right"):
                if side in proto:
                    self.rewrite([proto[side]])
        for key, val in proto.items():
            if key == "transform" and "transform" in entry:
                entry["transform"][0:0] = val
            elif key != "material" or "material" not in entry:
                entry[key] = val

    def _children(self, entry, name):
        if entry.get("add") == "group" and "children" in entry:
            self.rewrite(entry["children"])


def yaml_file_to_world_objects(file_path):
    with open(file_path, "r") as f:
        tree = yaml.load(f, Loader=_Loader)
    return yaml_tree_to_world_objects(tree)


def yaml_text_to_world_objects(text):
    return yaml_tree_to_world_objects(yaml.load(text, Loader=_Loader))


def yaml_tree_to_world_objects(tree):
    if tree is None:
        return []
    macros = _Macros(tree)
    macros.rewrite(tree)
    scene = {"camera": None, "lights": [], "world": [], "config": None}
    makers = {"camera": lambda e: scene.__setitem__("camera", Camera.from_yaml(e)),
              "light": lambda e: scene["lights"].append(Light.from_yaml(e)),
              "config": lambda e: scene.__setitem__("config", GlobalConfig.from_yaml(e))}
    for entry in tree:
        if "add" not in entry:
            continue
        make = makers.get(entry["add"])
        if make is not None:
            make(entry)
            continue
        shape = recursive_add(entry, macros.table)
        if shape is not None:
            scene["world"].append(shape)
    scene["world"] = [Group(material=Material(), transform=matrix4x4identity(), children=scene["world"])]
    return scene


def recursive_add(tree, defines):
    return Shape._recursive_helper(tree, defines)


def expand_defines_in_tree(tree, defines):
    """The reference's name for the rewrite pass (yaml_parser.py:70), kept for callers that use it directly."""
    m = _Macros([])
    m.table = defines
    m.rewrite(tree)
