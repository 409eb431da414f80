"""YAML scene files -> camera, lights, world (drop-in for the reference's `yaml_parser.py`).

`yaml_file_to_world_objects(path)` returns the same dict the reference does
(yaml_parser.py:10-64): `{'camera', 'lights', 'world': [root Group], 'config'}`.
The `define` / `extend` macro language is the reference's (yaml_parser.py:26-47, 70-134):
  * `define: name, value: ...` registers a constant; `extend: parent` first copies the
    parent dict and overlays the child's keys;
  * a define name inside a `value` / `transform` list is spliced in place;
  * `material: name` is replaced by the define, `material: {name: ..}`-style dicts overlay it;
  * `add: name` instantiates a defined shape: its keys are copied in, a material already on
    the instance wins, and the define's transform entries go BEFORE the instance's own.
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


def yaml_file_to_world_objects(file_path):
    with open(file_path, "r") as f:
        tree = yaml.load(f, Loader=_Loader)
    return yaml_tree_to_world_objects(tree)


def yaml_text_to_world_objects(text):
    return yaml_tree_to_world_objects(yaml.load(text, Loader=_Loader))


def yaml_tree_to_world_objects(tree):
    if tree is None:
        return []
    rv = {"camera": None, "lights": [], "world": [], "config": None}

    defines = {}
    extends = {}
    for obj in tree:
        if "define" in obj:
            defines[obj["define"]] = obj.get("value")
            if obj.get("extend") is not None:
                extends[obj["define"]] = obj["extend"]

    for name, parent_name in extends.items():
        merged = deepcopy(defines[parent_name])
        if type(merged) == dict:
            merged.update(defines[name])
            defines[name] = merged

    expand_defines_in_tree(tree, defines)

    for obj in tree:
        if "add" not in obj:
            continue
        kind = obj["add"]
        if kind == "camera":
            rv["camera"] = Camera.from_yaml(obj)
        elif kind == "light":
            rv["lights"].append(Light.from_yaml(obj))
        elif kind == "config":
            rv["config"] = GlobalConfig.from_yaml(obj)
        else:
            item = recursive_add(obj, defines)
            if item is not None:
                rv["world"].append(item)

    rv["world"] = [Group(material=Material(), transform=matrix4x4identity(), children=rv["world"])]
    return rv


def recursive_add(tree, defines):
    return Shape._recursive_helper(tree, defines)


def _splice(lst, name, items):
    i = lst.index(name)
    lst[i:i + 1] = list(items)


def expand_defines_in_tree(tree, defines):
    for obj in tree:
        for k in defines:
            if "value" in obj and k in obj["value"]:
                _splice(obj["value"], k, deepcopy(defines[k]))
            if "material" in obj and k in obj["material"]:
                if type(obj["material"]) == str:
                    obj["material"] = deepcopy(defines[k])
                elif type(obj["material"]) == dict:
                    overlay = obj["material"]
                    obj["material"] = deepcopy(defines[k])
                    obj["material"].update(overlay)
            if "transform" in obj and k in obj["transform"]:
                _splice(obj["transform"], k, deepcopy(defines[k]))
            if "extend" in obj and k in obj["extend"]:
                del obj["extend"]
                if "value" in obj and k in obj["value"]:
                    obj["value"][k] = deepcopy(defines[k])
                elif "value" in obj:
                    if type(obj["value"]) == dict:
                        overlay = obj["value"]
                        obj["value"] = deepcopy(defines[k])
                        obj["value"].update(overlay)
                    else:
                        obj["value"][0:0] = deepcopy(defines[k])
                else:
                    obj["value"] = deepcopy(defines[k])
            if "add" in obj:
                if k == obj["add"]:
                    inst = deepcopy(defines[k])
                    if inst.get("add") == "group" and "children" in inst:
                        expand_defines_in_tree(inst["children"], defines)
                    if inst.get("add") == "csg":
                        for side in ("left", "right"):
                            if side in inst:
                                expand_defines_in_tree([inst[side]], defines)
                    for key, val in inst.items():
                        if key == "material":
                            obj.setdefault("material", val)
                        elif key == "transform":
                            if "transform" not in obj:
                                obj["transform"] = val
                            else:
                                obj["transform"][0:0] = val
                        else:
                            obj[key] = val
                if obj["add"] == "group" and "children" in obj:
                    expand_defines_in_tree(obj["children"], defines)
