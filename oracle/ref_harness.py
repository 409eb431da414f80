"""TEST INFRASTRUCTURE ONLY, BUILD CONTAINER ONLY.  Imports the reference (Python, read-only
at /root/reference) to validate the oracle and to generate golden fixtures.  The reference
cannot travel to the GPU box, so nothing under tests/ marked `gpu`, smoke() or bench.py may
import this module."""
import contextlib
import os
import sys

REF = "/root/reference"
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stubs")


def available():
    return os.path.isdir(REF)


def import_reference():
    """-> dict of the reference's modules (renderer, shapes, canvas, matrix, vector, yaml_parser)."""
    if not available():
        raise RuntimeError("/root/reference is not present")
    for p in (REF, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib
    mods = {}
    for name in ("vector", "matrix", "canvas", "renderer", "shapes", "config", "obj_parser", "yaml_parser"):
        mods[name] = importlib.import_module(name)
    # The YAML path of the reference never passes `transform` to CubeMapPattern and raises
    # TypeError (shapes.py:1563-1569 vs 1843).  Harness-process-only default = identity, the only
    # value that call could have meant.
    sh = mods["shapes"]
    if not getattr(sh.CubeMapPattern, "_harness_patched", False):
        orig = sh.CubeMapPattern.__init__

        def patched(self, uv_left, uv_front, uv_right, uv_back, uv_up, uv_down, uv_map, transform=None):
            orig(self, uv_left, uv_front, uv_right, uv_back, uv_up, uv_down, uv_map,
                 mods["matrix"].matrix4x4identity() if transform is None else transform)
        sh.CubeMapPattern.__init__ = patched
        sh.CubeMapPattern._harness_patched = True
    return mods


@contextlib.contextmanager
def chdir(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


def load_reference_scene(yaml_path, cwd=None):
    """yaml_file_to_world_objects of the reference, run from `cwd` (relative asset paths)."""
    mods = import_reference()
    with chdir(cwd or os.getcwd()):
        objs = mods["yaml_parser"].yaml_file_to_world_objects(os.path.abspath(yaml_path) if cwd is None else yaml_path)
    w = mods["renderer"].World(objs["lights"], objs["world"])
    return mods, objs, w
