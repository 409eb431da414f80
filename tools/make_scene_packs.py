"""BUILD CONTAINER ONLY.  Reads the reference's scene inputs (YAML + OBJ + the one image it
ships) with THIS package's loaders and writes scene packs under scenes/, so that tests and
bench.py have the BASELINE.json scenes on machines without /root/reference.

Divergences from the files as shipped, all recorded in each pack's meta:
  * checkered_cube / skybox: CubeMapPattern gets an identity transform (the reference raises);
  * earth: `file:` points at the only image in the tree, texture_mapping/orrery/earthmap-small.jpg;
  * skybox: six synthetic 32x32 faces (numpy default_rng(0)), the originals are absent;
  * shadow_glamour_shot_nojitter: `jitter: false` (the reference's jitter is not reproducible).
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_harness import chdir  # noqa: E402  (a cwd helper only; the reference is not imported)
from ray_tracer_challenge_amd.renderer import World  # noqa: E402
from ray_tracer_challenge_amd.scene_io import save_scene  # noqa: E402
from ray_tracer_challenge_amd.yaml_parser import yaml_file_to_world_objects, yaml_text_to_world_objects  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")


def synthetic_faces(tmpdir):
    """Six 32x32 RGB cube faces: seeded gradient + checker."""
    from PIL import Image
    rng = np.random.default_rng(0)
    paths = {}
    for name in ("negx", "posx", "negy", "posy", "negz", "posz"):
        base = rng.integers(40, 255, size=3)
        yy, xx = np.mgrid[0:32, 0:32]
        img = np.zeros((32, 32, 3), np.uint8)
        for c in range(3):
            img[:, :, c] = (base[c] * (0.35 + 0.65 * ((xx // 4 + yy // 4) % 2)) * (0.5 + xx / 64.0)).clip(0, 255)
        img[0:4, 0:4] = rng.integers(0, 255, size=3)
        path = os.path.join(tmpdir, name + ".png")
        Image.fromarray(img).save(path)
        paths[name] = path
    return paths


def pack(name, yaml_path, cwd, meta=None, edit=None):
    with chdir(cwd):
        if edit is None:
            objs = yaml_file_to_world_objects(yaml_path)
        else:
            with open(yaml_path) as f:
                objs = yaml_text_to_world_objects(edit(f.read()))
    m = {"source": os.path.relpath(yaml_path, REF) if os.path.isabs(yaml_path) else yaml_path}
    m.update(meta or {})
    out = os.path.join(OUT, name + ".npz")
    save_scene(out, objs["camera"], World(objs["lights"], objs["world"]), objs["config"], m)
    print("%-32s %8.1f KB" % (name, os.path.getsize(out) / 1024.0))


def main(which=()):
    os.makedirs(OUT, exist_ok=True)
    # a scene of this repository's own (scenes/src/): perturbed patterns, which no scene file of the reference uses
    if not which or "perturbed" in which:
        pack("perturbed", os.path.join(OUT, "src", "perturbed.yml"), OUT,
             {"source": "scenes/src/perturbed.yml (this repository)", "note": "noise = restated pnoise3, parity unpinned"})
    # another: the solid patterns and composites no scene file of the reference uses, pinned by a reference-rendered fixture
    if not which or "patterns" in which:
        pack("patterns", os.path.join(OUT, "src", "patterns.yml"), OUT, {"source": "scenes/src/patterns.yml (this repository)"})
    if not which or "uvmaps" in which:
        pack("uvmaps", os.path.join(OUT, "src", "uvmaps.yml"), REF,
             {"source": "scenes/src/uvmaps.yml (this repository)", "divergence": "CubeMapPattern transform = identity (reference raises TypeError)"})
    if which and "reference" not in which:
        return
    tm = os.path.join(REF, "texture_mapping")
    pack("cover", os.path.join(REF, "cover/cover.yml"), REF)
    pack("csg_test", os.path.join(REF, "csg_test.yml"), REF)
    pack("group", os.path.join(REF, "group_test/group.yml"), REF)
    pack("shadow_glamour_shot", os.path.join(REF, "shadow_glamour_shot/shadow_glamour_shot.yml"), REF)
    pack("shadow_glamour_shot_nojitter", os.path.join(REF, "shadow_glamour_shot/shadow_glamour_shot.yml"), REF,
         {"divergence": "jitter: false"}, lambda t: t.replace("jitter: true", "jitter: false"))
    for n in ("checkered_sphere", "checkered_plane", "checkered_cylinder", "align_check_plane"):
        pack(n, os.path.join(tm, n, n + ".yml"), REF)
    pack("checkered_cube", os.path.join(tm, "checkered_cube/checkered_cube.yml"), REF,
         {"divergence": "CubeMapPattern transform = identity (reference raises TypeError)"})
    pack("earth", os.path.join(tm, "earth/earth.yml"), REF,
         {"divergence": "texture = texture_mapping/orrery/earthmap-small.jpg (earthmap1k.jpg is absent)"},
         lambda t: t.replace("texture_mapping/earth/earthmap1k.jpg", "texture_mapping/orrery/earthmap-small.jpg"))
    with tempfile.TemporaryDirectory() as tmp:
        faces = synthetic_faces(tmp)

        def sky(t):
            for k, p in faces.items():
                t = t.replace("texture_mapping/skybox/%s.jpg" % k, p)
            return t
        pack("skybox", os.path.join(tm, "skybox/skybox.yml"), REF,
             {"divergence": "six synthetic 32x32 faces (default_rng(0)); CubeMapPattern transform = identity"}, sky)
    pack("teapot", os.path.join(REF, "obj_file_test/obj_file_test.yml"), os.path.join(REF, "objects"))
    pack("dragons", os.path.join(REF, "bounding_box_images/bounding_boxes.yml"), os.path.join(REF, "objects"))


if __name__ == "__main__":
    main(sys.argv[1:])          # no argument: every pack; `perturbed`: only this repository's own scene
