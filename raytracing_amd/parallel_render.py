"""Tile-sharded render across the GPUs of one node: one process per GPU (torch.distributed, backend
nccl = RCCL over xGMI), the scene replicated on every GPU, tile t rendered by rank t % world (tile rows
are rotated against each other, see tile_column, so a rank's tiles form diagonals), the
tile buffers gathered on rank 0 and scattered into the row-major image.  The only exchange step is
that gather (<= 25 MB for a 1080p float image in total); there is no data-path collective during the
render itself.

The render call is injected (`render_tiles_fn`) so the sharding / gather / scatter logic can be
exercised on CPU with the gloo backend; on a GPU it is api.DeviceScene.render_device.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import api


class TileLayout:
    """Bookkeeping of the tile split (mirrors rt_tiles_per_rank / rt_tiles_to_image)."""

    def __init__(self, w, h, tile_w, tile_h, world):
        self.w, self.h, self.tw, self.th, self.world = w, h, tile_w, tile_h, world
        self.tiles_x = (w + tile_w - 1) // tile_w
        self.tiles_y = (h + tile_h - 1) // tile_h
        self.n_tiles = self.tiles_x * self.tiles_y
        self.per_rank = (self.n_tiles + world - 1) // world  # rank 0 holds the most tiles
        self.tile_floats = tile_w * tile_h * 3

    def tile_column(self, t):
        """Column of tile t (rt_tile_column): row t // tiles_x is rotated by its own index."""
        return (t % self.tiles_x + t // self.tiles_x) % self.tiles_x

    def tile_origin(self, t):
        return self.tile_column(t) * self.tw, (t // self.tiles_x) * self.th

    def local_tiles(self, rank):
        return (self.n_tiles + self.world - 1 - rank) // self.world

    def buffer_floats(self):
        """Every rank allocates the same (maximum) size so the gather is regular."""
        return self.per_rank * self.tile_floats

    def scatter_index(self):
        """For every image pixel (row-major) the index of its RGB triple in the rank-major gathered buffer."""
        ys, xs = np.mgrid[0:self.h, 0:self.w]
        ty = ys // self.th
        t = ty * self.tiles_x + (xs // self.tw - ty) % self.tiles_x  # inverse of tile_column
        src = ((t % self.world) * self.per_rank + t // self.world) * (self.tw * self.th) + (ys % self.th) * self.tw + (xs % self.tw)
        return src.ravel().astype(np.int64)


class ShardedRenderer:
    """rank / world from torch.distributed (or 0 / 1 when it is not initialised).

    step() renders frame i into one of two tile buffers and starts its gather without waiting for it, then
    finishes frame i-1 (waits for that gather, scatters it to the row-major image on rank 0): the exchange of
    one frame runs while the next one renders.  flush() finishes the frame still in flight.  With
    pipelined=False every step() finishes its own frame."""

    def __init__(self, w, h, tile_w, tile_h, device, render_tiles_fn, pipelined=True, native_gather=False):
        """native_gather: the exchange is rt_gather_tiles_device (RCCL point-to-point calls made by the library, enqueued on
        the current stream; needs api.comm_init) instead of torch.distributed.gather."""
        self.native_gather = native_gather
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.layout = TileLayout(w, h, tile_w, tile_h, self.world)
        self.device = device
        self.render_tiles_fn = render_tiles_fn
        self.pipelined = pipelined and self.world > 1
        nbuf = 2 if self.pipelined else 1
        self.mine = [torch.zeros(self.layout.buffer_floats(), dtype=torch.float32, device=device) for _ in range(nbuf)]
        self.gathered_flat = [None] * nbuf
        self.gathered = [None] * nbuf
        self.image = None
        self.perm = None
        self.frame = 0
        self.pending = None  # (work, buffer index) of the frame whose gather is in flight
        if self.rank == 0:
            if self.world > 1:
                # one block per buffer, rank-major: the gather writes into its slices, no concatenation afterwards
                for k in range(nbuf):
                    self.gathered_flat[k] = torch.empty(self.world * self.mine[k].numel(), dtype=torch.float32, device=device)
                    self.gathered[k] = list(self.gathered_flat[k].chunk(self.world))
            self.image = torch.zeros(h * w * 3, dtype=torch.float32, device=device)
            self.perm = torch.from_numpy(self.layout.scatter_index()).to(device)
        if self.world > 1 and not self.native_gather:
            # set-up, not a frame: the first gather opens the point-to-point channels to rank 0 (tens of ms on RCCL)
            probe = torch.zeros(1, dtype=torch.float32, device=device)
            dist.gather(probe, [torch.zeros_like(probe) for _ in range(self.world)] if self.rank == 0 else None, dst=0)

    def _finish(self, pending):
        """Wait (on the current stream) for a frame's gather and scatter it; the image view on rank 0, else None."""
        work, k = pending
        if work is not None:
            work.wait()
        if self.rank != 0:
            return None
        src = self.gathered_flat[k] if self.world > 1 else self.mine[k]
        torch.index_select(src.view(-1, 3), 0, self.perm, out=self.image.view(-1, 3))
        return self.image.view(self.layout.h, self.layout.w, 3)

    def step(self):
        """Render this rank's tiles of the next frame and start its gather.  Returns the image (h, w, 3) of the
        frame finished by this call on rank 0 -- the previous frame when pipelined, None for the first call --
        and None on the other ranks.  Everything is enqueued on the current stream."""
        k = self.frame % len(self.mine)
        self.frame += 1
        self.render_tiles_fn(self.mine[k], self.rank, self.world)
        work = None
        if self.world > 1 and self.native_gather:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            api.gather_tiles_device(self.mine[k].data_ptr(), self.mine[k].numel(), self.gathered_flat[k].data_ptr() if self.rank == 0 else 0, 0, stream)
        elif self.world > 1:
            work = dist.gather(self.mine[k], self.gathered[k], dst=0, async_op=True)
        if not self.pipelined:
            return self._finish((work, k))
        prev, self.pending = self.pending, (work, k)
        # finishing frame i-1 here also keeps frame i+1 from overwriting a buffer whose gather still runs
        return self._finish(prev) if prev is not None else None

    def flush(self):
        """Finish the frame still in flight (pipelined mode); returns its image on rank 0."""
        prev, self.pending = self.pending, None
        return self._finish(prev) if prev is not None else None


def device_render_fn(scene, cam, w, h, spp, seed, tile_w, tile_h, counters=None):
    """render_tiles_fn for a real GPU: rt_render_device on torch's current stream."""
    def fn(out, rank, world):
        stream = torch.cuda.current_stream(out.device)
        scene.render_device(cam, w, h, spp, seed, tile_w, tile_h, rank, world, out.data_ptr(),
                            counters.data_ptr() if counters is not None else 0, stream.cuda_stream)
    return fn


def new_counters(device):
    return torch.zeros(api.COUNTER_SLOTS, api.COUNTER_STRIDE, dtype=torch.int64, device=device)


class ProgressiveRenderer:
    """Progressive accumulation over the same tile split: every pass adds spp_pass samples to the running sums
    that stay in this rank's tile buffer (rt_render_pass_device semantics: the pass that reaches spp_total
    divides by it), and nothing is exchanged until display() gathers the buffers on rank 0.

    render_pass_fn(out, rank, world, sample0, spp_pass, spp_total) renders one pass into `out`."""

    def __init__(self, w, h, tile_w, tile_h, device, render_pass_fn, spp_total):
        self.sharded = ShardedRenderer(w, h, tile_w, tile_h, device, None, pipelined=False)
        self.render_pass_fn = render_pass_fn
        self.spp_total = spp_total
        self.done = 0

    @property
    def rank(self):
        return self.sharded.rank

    def add_pass(self, spp_pass):
        """Samples [done, done + spp_pass) of every pixel of this rank's tiles; enqueued on the current stream."""
        if spp_pass <= 0 or self.done + spp_pass > self.spp_total:
            raise ValueError("pass outside the total sample count")
        s = self.sharded
        self.render_pass_fn(s.mine[0], s.rank, s.world, self.done, spp_pass, self.spp_total)
        self.done += spp_pass

    def display(self):
        """Gather the tile buffers and return the image so far on rank 0 (None elsewhere): the running sums
        divided by the samples taken, or the finished image once all samples are in."""
        s = self.sharded
        work = dist.gather(s.mine[0], s.gathered[0], dst=0, async_op=True) if s.world > 1 else None
        img = s._finish((work, 0))
        if img is None or self.done == 0 or self.done == self.spp_total:
            return img
        return img / float(self.done)


def device_pass_fn(scene, cam, w, h, seed, tile_w, tile_h, counters=None):
    """render_pass_fn for a real GPU: rt_render_pass_device on torch's current stream."""
    def fn(out, rank, world, sample0, spp_pass, spp_total):
        stream = torch.cuda.current_stream(out.device)
        scene.render_pass_device(cam, w, h, spp_pass, sample0, spp_total, seed, tile_w, tile_h, rank, world, out.data_ptr(),
                                 counters.data_ptr() if counters is not None else 0, stream.cuda_stream)
    return fn
