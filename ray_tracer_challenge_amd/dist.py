"""Multi-GPU frame rendering: one process per GPU, image tiles sharded, one gather.

The reference farms 2x2-pixel tiles to a process pool that writes a shared-memory canvas
(renderer.py:781-835, canvas.py:60-61).  Here the unit is a 32x32 tile; tile i belongs to rank
i mod N (interleaved, because sky and mesh pixels differ in cost by orders of magnitude); every
rank holds the whole compiled scene; each traces its tiles into a compact [slot][ty][tx][c]
buffer; ONE collective -- a gather to rank 0 (RCCL over xGMI with the `nccl` backend, `gloo`
in CPU tests) -- and rank 0 un-permutes the tiles into the [x][y][c] canvas.  No rays or
partial results cross GPUs, so there is nothing else to exchange.
"""
import numpy as np

TILE = 32


def tiles_for(hsize, vsize, tile=TILE):
    return (hsize + tile - 1) // tile, (vsize + tile - 1) // tile


def rank_tiles(hsize, vsize, world, rank, tile=TILE):
    """Tile ids of `rank` (row-major tile numbering, interleaved over ranks)."""
    tx, ty = tiles_for(hsize, vsize, tile)
    return np.arange(tx * ty, dtype=np.int32)[rank::world]


def slots_per_rank(hsize, vsize, world, tile=TILE):
    tx, ty = tiles_for(hsize, vsize, tile)
    return (tx * ty + world - 1) // world


def unpermute_index(hsize, vsize, world, tile=TILE):
    """(src, dst): flat pixel indices such that canvas.view(-1, 3)[dst] = gathered.view(-1, 3)[src],
    where `gathered` is the concatenation over ranks of their (slots_per_rank, T, T, 3) buffers."""
    tx, _ = tiles_for(hsize, vsize, tile)
    per = slots_per_rank(hsize, vsize, world, tile)
    slots = np.concatenate([np.pad(rank_tiles(hsize, vsize, world, r, tile),
                                   (0, per - len(rank_tiles(hsize, vsize, world, r, tile))), constant_values=-1)
                            for r in range(world)])
    ly, lx = np.meshgrid(np.arange(tile), np.arange(tile), indexing="ij")
    gx = (slots[:, None, None] % tx) * tile + lx[None]
    gy = (slots[:, None, None] // tx) * tile + ly[None]
    ok = (slots[:, None, None] >= 0) & (gx < hsize) & (gy < vsize)
    return np.nonzero(ok.ravel())[0].astype(np.int64), (gx * vsize + gy)[ok].astype(np.int64)


class FrameGather(object):
    """Preallocated buffers + tile tables for gathering one frame per call to rank 0.

    On a GPU the un-permute is one HIP kernel (`rtc_scatter_tiles_device`) over the gathered buffer, which
    is a single allocation whose per-rank slices the collective receives into; on the CPU (gloo tests)
    it is a torch index copy with the same tile tables."""

    def __init__(self, hsize, vsize, world, rank, device, tile=TILE, group=None, with_ids=False):
        import torch
        self.torch = torch
        self.hsize, self.vsize, self.world, self.rank, self.tile, self.group = hsize, vsize, world, rank, tile, group
        self.with_ids = with_ids        # also gather the hit-id tiles (a second, 6 x smaller collective) into `self.ids`
        self.mine = rank_tiles(hsize, vsize, world, rank, tile)
        self.per = per = slots_per_rank(hsize, vsize, world, tile)
        self.on_gpu = torch.device(device).type == "cuda"
        self.tiles_dev = torch.from_numpy(self.mine.copy()).to(device)
        self.frame = None
        self.gathered = None
        if rank == 0:
            self.frame = torch.zeros((hsize, vsize, 3), dtype=torch.float64, device=device)
            # every rank's slots back to back: the collective receives straight into the slices
            self.all_tiles = torch.zeros((world, per, tile, tile, 3), dtype=torch.float64, device=device)
            self.compact = self.all_tiles[0] if world == 1 else torch.zeros((per, tile, tile, 3), dtype=torch.float64, device=device)
            self.gathered = [self.all_tiles[r] for r in range(world)]
            slots = np.full((world, per), -1, np.int32)
            for r in range(world):
                t = rank_tiles(hsize, vsize, world, r, tile)
                slots[r, :len(t)] = t
            self.slot_tiles = torch.from_numpy(slots.ravel().copy()).to(device)
            if not self.on_gpu:
                src, dst = unpermute_index(hsize, vsize, world, tile)
                self.src, self.dst = torch.from_numpy(src), torch.from_numpy(dst)
        else:
            self.compact = torch.zeros((per, tile, tile, 3), dtype=torch.float64, device=device)
        self.ids = self.compact_ids = self.all_ids = self.gathered_ids = None
        if with_ids:
            if rank == 0:
                self.ids = torch.full((hsize, vsize), -1, dtype=torch.int32, device=device)
                self.all_ids = torch.full((world, per, tile, tile), -1, dtype=torch.int32, device=device)
                self.compact_ids = self.all_ids[0] if world == 1 else torch.full((per, tile, tile), -1, dtype=torch.int32, device=device)
                self.gathered_ids = [self.all_ids[r] for r in range(world)]
            else:
                self.compact_ids = torch.full((per, tile, tile), -1, dtype=torch.int32, device=device)

    def scatter(self):
        """rank 0: the gathered tiles -> the canvas."""
        torch = self.torch
        if self.on_gpu:
            from . import _ffi
            _ffi.scatter_tiles_device(self.all_tiles.data_ptr(), self.slot_tiles.data_ptr(), self.world * self.per, self.tile,
                                      self.hsize, self.vsize, self.frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
        else:
            flat = self.all_tiles.reshape(-1, 3)
            self.frame.view(-1, 3).index_copy_(0, self.dst, flat.index_select(0, self.src))
        if self.with_ids:
            if self.on_gpu:
                from . import _ffi
                _ffi.scatter_id_tiles_device(self.all_ids.data_ptr(), self.slot_tiles.data_ptr(), self.world * self.per, self.tile,
                                             self.hsize, self.vsize, self.ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
            else:
                self.ids.view(-1).index_copy_(0, self.dst, self.all_ids.reshape(-1).index_select(0, self.src))

    def gather(self):
        """After this rank's tiles are in `self.compact`: returns the canvas tensor on rank 0, None elsewhere."""
        torch = self.torch
        if self.world > 1:
            import torch.distributed as dist
            dist.gather(self.compact, self.gathered if self.rank == 0 else None, dst=0, group=self.group)
            if self.with_ids:
                dist.gather(self.compact_ids, self.gathered_ids if self.rank == 0 else None, dst=0, group=self.group)
            if self.rank != 0:
                return None
        self.scatter()
        return self.frame


class FramePipeline(object):
    """Frame k's gather overlapped with frame k+1's tracing (a stream of frames: animation, bench).

    Two FrameGather buffer sets alternate.  `submit()` starts the gather of the set just traced as an
    asynchronous collective (on RCCL's own stream, ordered after the trace kernel) and returns at once;
    the set is collected -- wait for its gather, tiles -> canvas on rank 0 -- right before it is traced
    into again, or by `drain()`.  Every frame is whole when `drain()` returns; what changes is that the
    xGMI transfer no longer sits between two frames' kernels."""

    def __init__(self, hsize, vsize, world, rank, device, tile=TILE, group=None, with_ids=False):
        self.sets = [FrameGather(hsize, vsize, world, rank, device, tile, group, with_ids) for _ in range(2)]
        self.pending = [None, None]
        self.k = 0
        self.world, self.rank = world, rank
        self.sets[1].tiles_dev = self.sets[0].tiles_dev      # one tile list: the scene's launch-order feedback is keyed by it
        if rank == 0:                       # one canvas: frames are assembled in order
            self.sets[1].frame = self.sets[0].frame
            self.sets[1].ids = self.sets[0].ids

    @property
    def frame(self):
        return self.sets[0].frame

    def current(self):
        """The buffer set to trace the next frame into (collected first if its last gather is still out)."""
        b = self.k % 2
        self._collect(b)
        return self.sets[b]

    def submit(self):
        """The current set has been traced into: start its gather."""
        b = self.k % 2
        fg = self.sets[b]
        if self.world > 1:
            import torch.distributed as dist
            work = [dist.gather(fg.compact, fg.gathered if self.rank == 0 else None, dst=0, group=fg.group, async_op=True)]
            if fg.with_ids:
                work.append(dist.gather(fg.compact_ids, fg.gathered_ids if self.rank == 0 else None, dst=0, group=fg.group, async_op=True))
            self.pending[b] = work
        else:
            self.pending[b] = True
        self.k += 1

    def _collect(self, b):
        work = self.pending[b]
        if work is None:
            return
        self.pending[b] = None
        if work is not True:
            for w in work:
                w.wait()
        if self.rank == 0:
            self.sets[b].scatter()

    def drain(self):
        """-> the canvas of the last frame submitted (rank 0), all gathers finished."""
        first = self.k % 2                  # older set first
        self._collect(first)
        self._collect(1 - first)
        return self.frame if self.rank == 0 else None


def reduce_counters(rays, group=None):
    """The 13 ray / work counters every rank's kernels added to (`rays_dev` of rtc_render_device: a torch int64 tensor)
    summed onto rank 0 -- the `ncclReduce` of SURVEY.md 8e.  In place; other ranks' tensors are left as they were."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(rays, dst=0, op=dist.ReduceOp.SUM, group=group)
    return rays


def render_distributed(scene, cam, fg, precision="fp64", max_depth=5, rays_ptr=None, stream=None, count_work=False):
    """One frame: trace this rank's tiles with the HIP kernel into fg.compact (and fg.compact_ids if the FrameGather
    was made with_ids), then gather.  `scene` is an `_ffi.Scene` on this rank's GPU; `fg` a FrameGather built for
    `cam`'s size.  Ray counters stay per rank until `reduce_counters`."""
    scene.render_device(cam, fg.compact.data_ptr(), ids_ptr=fg.compact_ids.data_ptr() if fg.with_ids else None,
                        rays_ptr=rays_ptr, tile_ids_ptr=fg.tiles_dev.data_ptr(),
                        n_tiles=len(fg.mine), precision=precision, max_depth=max_depth, tile_size=fg.tile,
                        stream=stream, count_work=count_work)
    return fg.gather()


def render_pipelined(scene, cam, pipe, precision="fp64", max_depth=5, rays_ptr=None, stream=None, count_work=False):
    """One frame of a stream of frames: trace this rank's tiles into the pipeline's current buffer set and start its
    gather; the frame is complete after `pipe.drain()` (or once its set comes round again)."""
    fg = pipe.current()
    scene.render_device(cam, fg.compact.data_ptr(), ids_ptr=fg.compact_ids.data_ptr() if fg.with_ids else None,
                        rays_ptr=rays_ptr, tile_ids_ptr=fg.tiles_dev.data_ptr(),
                        n_tiles=len(fg.mine), precision=precision, max_depth=max_depth, tile_size=fg.tile,
                        stream=stream, count_work=count_work)
    pipe.submit()
