"""The C-ABI library: loads without a GPU, exports every symbol include/rtc_api.h declares,
and fails loudly (never silently falls back) when there is no device."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_pack
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rtc_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtc_[a-z_]+)\s*\(", text)))


def test_exports_match_header():
    names = declared_functions()
    assert sorted(_ffi.EXPORTS) == names
    L = _ffi.lib()
    for n in names:
        assert getattr(L, n) is not None


def test_version_and_device_count():
    assert _ffi.lib().rtc_version() == 100
    assert _ffi.device_count() >= 0


def test_struct_layout_matches_header():
    # field order of rtc_scene_desc in the header == _DESC_FIELDS in the binding
    text = open(os.path.join(ROOT, "include", "rtc_api.h")).read()
    body = text[text.index("typedef struct {", text.index("Flattened scene")):text.index("} rtc_scene_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S).replace("typedef struct {", "")
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        names = decl.split(",")
        first = names[0].split()[-1]
        fields.append(first.lstrip("*"))
        fields.extend(n.strip().lstrip("*") for n in names[1:])
    assert fields == [n for n, _ in _ffi._DESC_FIELDS]


@pytest.mark.skipif(_ffi.device_count() > 0, reason="a GPU is present")
def test_no_gpu_is_an_error_not_a_fallback():
    cam, world, _ = load_pack("cover", (16, 16))
    cs = compile_scene(world)
    with pytest.raises(_ffi.RtcError, match="no GPU"):
        _ffi.Scene(cs)
    from ray_tracer_challenge_amd.renderer import render, render_multi
    with pytest.raises(_ffi.RtcError):
        render(cam, world)
    with pytest.raises(RuntimeError):
        render_multi(cam, world)


def test_validation_rejects_bad_scene():
    cam, world, _ = load_pack("cover")
    cs = compile_scene(world)
    cs.arrays["leaf_material"] = cs.arrays["leaf_material"].copy()
    cs.arrays["leaf_material"][0] = 10 ** 6
    with pytest.raises(_ffi.RtcError, match="material out of range"):
        _ffi.Scene(cs)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "ray_tracer_challenge_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "rtc_oracle" not in src, f
