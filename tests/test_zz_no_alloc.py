"""rtc_render_device_noalloc: the variant of rtc_render_device whose launch-order buffers (wave costs, order lists) live
in a device array inside the library instead of a workspace the handle allocates -- for callers that cannot call
hipMalloc.  Through the C ABI, against rtc_render_device on the same inputs: bit-identical frames, hit ids and ray
counts, on several calls in a row (plain order, sorted order, kept order), then on a SMALLER frame (the array still holds
the larger frame's costs and order), then on another scene; and the size limit is refused with a clear error."""
import numpy as np
import pytest

from conftest import load_pack
from ray_tracer_challenge_amd import _ffi
from ray_tracer_challenge_amd.compile import compile_scene

pytestmark = pytest.mark.gpu


def _both(sc, cam, precision, torch, **kw):
    dev = torch.device("cuda", 0)
    out = []
    for noalloc in (False, True):
        frame = torch.full((cam.hsize, cam.vsize, 3), -7.0, dtype=torch.float64, device=dev)
        ids = torch.full((cam.hsize, cam.vsize), -7, dtype=torch.int32, device=dev)
        rays = torch.zeros(13, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        sc.render_device(cam, frame.data_ptr(), ids_ptr=ids.data_ptr(), rays_ptr=rays.data_ptr(), precision=precision, stream=stream,
                         noalloc=noalloc, **kw)
        torch.cuda.synchronize()
        out.append((frame.cpu().numpy(), ids.cpu().numpy(), rays.cpu().numpy()))
    return out


def test_noalloc_entry_point_matches_render_device():
    import torch
    cam, world, _ = load_pack("cover")
    with _ffi.Scene(compile_scene(world), device=0) as sc:
        big = cam.resized(200, 120)
        small = cam.resized(72, 40)
        for view in (big, big, big, small, small, big):          # two calls in a row and more; the second size is smaller
            for precision in ("fp64", "fp32"):
                (a, ia, ra), (b, ib, rb) = _both(sc, view, precision, torch)
                np.testing.assert_array_equal(a, b)
                np.testing.assert_array_equal(ia, ib)
                np.testing.assert_array_equal(ra, rb)
                assert ra[0] == view.hsize * view.vsize and (a != -7.0).all()
        (a, ia, ra), (b, ib, rb) = _both(sc, small, "fp64", torch, feedback=False)
        np.testing.assert_array_equal(a, b)
        # beyond the array's size: refused, nothing launched
        huge = cam.resized(4096, 1032)
        frame = torch.zeros(1, dtype=torch.float64, device="cuda:0")
        with pytest.raises(_ffi.RtcError, match="at most"):
            sc.render_device(huge, frame.data_ptr(), noalloc=True)
    cam2, world2, _ = load_pack("teapot")
    with _ffi.Scene(compile_scene(world2), device=0) as sc2:                     # another scene on the same GPU: same array
        for _ in range(3):
            (a, ia, ra), (b, ib, rb) = _both(sc2, cam2, "fp64", torch)
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(ia, ib)
            np.testing.assert_array_equal(ra, rb)
