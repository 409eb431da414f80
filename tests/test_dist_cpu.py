"""The multi-GPU path's host logic with world_size 2 on the CPU (gloo): tile partition, the one
gather, the un-permute.  Each rank's tiles are filled by the CPU oracle in place of the HIP
kernel (this is a test: the product path never does that), so the assembled frame can be
compared with a whole-frame render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pack
from ray_tracer_challenge_amd.dist import FrameGather, rank_tiles, slots_per_rank, tiles_for, unpermute_index


def test_partition_covers_every_tile_once():
    for (w, h, n) in [(100, 100, 2), (1920, 1080, 8), (33, 65, 3), (8, 8, 4)]:
        tx, ty = tiles_for(w, h)
        got = np.sort(np.concatenate([rank_tiles(w, h, n, r) for r in range(n)]))
        assert np.array_equal(got, np.arange(tx * ty))
        src, dst = unpermute_index(w, h, n)
        assert len(dst) == w * h and len(np.unique(dst)) == w * h
        assert src.max() < n * slots_per_rank(w, h, n) * 32 * 32


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import Oracle
        cam, scene_world, _ = load_pack("cover", (72, 40))
        fg = FrameGather(cam.hsize, cam.vsize, world, rank, torch.device("cpu"))
        tx, _ = tiles_for(cam.hsize, cam.vsize)
        T = fg.tile
        o = Oracle(scene_world)
        buf = fg.compact.numpy()
        for k, tile in enumerate(fg.mine):          # what rtc_render_device writes: [slot][ty][tx][c]
            x0, y0 = (tile % tx) * T, (tile // tx) * T
            xs, ys = np.meshgrid(np.arange(x0, min(x0 + T, cam.hsize)), np.arange(y0, min(y0 + T, cam.vsize)), indexing="xy")
            rgb, _, _, _ = o.render_pixels(cam, xs.ravel(), ys.ravel(), threads=2)
            buf[k, :xs.shape[0], :xs.shape[1]] = rgb.reshape(xs.shape[0], xs.shape[1], 3)
        frame = fg.gather()
        assert (frame is None) == (rank != 0)
        if rank == 0:
            np.save(out_path, frame.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame(tmp_path, oracle_cls):
    out = str(tmp_path / "frame.npy")
    port = 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    cam, world, _ = load_pack("cover", (72, 40))
    want, _, _ = oracle_cls(world).render(cam)
    np.testing.assert_array_equal(got, want)


def _fill(fg, cam_w, cam_h, frame_no):
    """What a trace kernel would write for an image whose pixel (x, y) has the value frame_no * 1e6 + x * h + y."""
    tx, _ = tiles_for(cam_w, cam_h)
    T = fg.tile
    buf = fg.compact.numpy()
    buf[:] = -1.0
    for k, tile in enumerate(fg.mine):
        x0, y0 = (tile % tx) * T, (tile // tx) * T
        for ly in range(min(T, cam_h - y0)):
            xs = np.arange(x0, min(x0 + T, cam_w))
            buf[k, ly, :len(xs), :] = (frame_no * 1e6 + xs * cam_h + (y0 + ly))[:, None]


def _pipeline_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ray_tracer_challenge_amd.dist import FramePipeline
        w, h = 70, 45
        pipe = FramePipeline(w, h, world, rank, torch.device("cpu"))
        seen = []
        for frame_no in range(5):
            fg = pipe.current()                 # collects the frame traced into this set two steps ago
            if rank == 0 and frame_no >= 2:
                seen.append(pipe.frame.numpy().copy())           # canvas now holds frame_no - 2
            _fill(fg, w, h, frame_no)
            pipe.submit()
        last = pipe.drain()
        assert (last is None) == (rank != 0)
        if rank == 0:
            seen.append(last.numpy().copy())
            np.save(out_path, np.stack(seen))
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_keeps_frames_whole_and_in_order(tmp_path):
    out = str(tmp_path / "frames.npy")
    port = 31000 + os.getpid() % 2000
    mp.spawn(_pipeline_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)                          # canvases after collecting frames 0, 1, 2 and, after drain, 4
    w, h = 70, 45
    base = (np.arange(w)[:, None] * h + np.arange(h)[None, :]).astype(np.float64)
    for canvas, frame_no in zip(got, (0, 1, 2, 4)):
        np.testing.assert_array_equal(canvas, np.repeat((frame_no * 1e6 + base)[:, :, None], 3, axis=2))


def _full_hd_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ray_tracer_challenge_amd.dist import FrameGather, reduce_counters
        w, h = 1920, 1080
        fg = FrameGather(w, h, world, rank, torch.device("cpu"), with_ids=True)
        tx, _ = tiles_for(w, h)
        T = fg.tile
        # what rtc_render_device writes for a frame whose pixel (x, y) has colour (v, v + 0.25, v + 0.5), v = x * h + y, and hit id v:
        # [slot][ty][tx] tiles, pixels outside the image untouched
        tiles = fg.mine.astype(np.int64)
        lx, ly = np.arange(T)[None, None, :], np.arange(T)[None, :, None]
        gx, gy = (tiles % tx)[:, None, None] * T + lx, (tiles // tx)[:, None, None] * T + ly
        inside = (gx < w) & (gy < h)
        v = (gx * h + gy).astype(np.float64)
        buf, ids = fg.compact.numpy(), fg.compact_ids.numpy()
        buf[:] = -1.0
        ids[:] = -5
        n = len(tiles)
        for c in range(3):
            buf[:n, :, :, c] = np.where(inside, v + 0.25 * c, -1.0)
        ids[:n] = np.where(inside, v, -5).astype(np.int32)
        counters = torch.tensor([n * T * T, rank + 1, 2 * rank, 7] + [0] * 8 + [rank * rank], dtype=torch.int64)
        frame = fg.gather()
        reduce_counters(counters)
        assert (frame is None) == (rank != 0)
        if rank == 0:
            np.savez(out_path, frame=frame.numpy(), ids=fg.ids.numpy(), counters=counters.numpy())
    finally:
        dist.destroy_process_group()


def test_eight_ranks_reassemble_a_1080p_frame_with_ids_and_counters(tmp_path):
    """The driver's scaling run in miniature, on the CPU: world_size 8 (gloo), the benched 1920x1080 frame -- 2 040 tiles
    of 32x32, the bottom row of tiles ragged (1080 = 33.75 tiles) -- partition, the colour gather, the hit-id gather,
    the un-permute, and the ray counters reduced onto rank 0."""
    out = str(tmp_path / "hd.npz")
    port = 33000 + os.getpid() % 2000
    world = 8
    mp.spawn(_full_hd_worker, args=(world, port, out), nprocs=world, join=True)
    z = np.load(out)
    w, h = 1920, 1080
    base = (np.arange(w)[:, None] * h + np.arange(h)[None, :]).astype(np.float64)
    for c in range(3):
        np.testing.assert_array_equal(z["frame"][:, :, c], base + 0.25 * c)
    np.testing.assert_array_equal(z["ids"], base.astype(np.int32))
    tx, ty = tiles_for(w, h)
    assert z["counters"][0] == tx * ty * 32 * 32                      # every tile traced exactly once
    assert z["counters"][1] == sum(r + 1 for r in range(world)) and z["counters"][2] == sum(2 * r for r in range(world))
    assert z["counters"][3] == 7 * world and z["counters"][12] == sum(r * r for r in range(world))
