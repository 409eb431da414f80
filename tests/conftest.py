import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_pack(name, size=None):
    from ray_tracer_challenge_amd.scene_io import load_world
    cam, world, objs = load_world(os.path.join(SCENES, name + ".npz"))
    if size is not None:
        cam = cam.resized(*size)
    return cam, world, objs


def load_golden(name):
    import json
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    g["meta"] = json.loads(g["meta"].tobytes().decode())
    return g


def ppm_levels(rgb):
    """construct_ppm's 8-bit levels (default clamping) of an (n, 3) or (w, h, 3) colour array."""
    from ray_tracer_challenge_amd.canvas import quantize
    a = np.asarray(rgb, dtype=np.float64)
    if a.ndim == 2:
        a = a.reshape(1, -1, 3)
    return quantize(a, True).astype(np.int32)


def frac_off_by_more_than_one(a, b):
    d = np.abs(ppm_levels(a) - ppm_levels(b)).max(axis=-1)
    return float((d > 1).mean())


@pytest.fixture(scope="session")
def oracle_cls():
    from oracle.oracle import Oracle, build
    build()
    return Oracle
