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
    """rank / world from torch.distributed (or 0 / 1 when it is not initialised)."""

    def __init__(self, w, h, tile_w, tile_h, device, render_tiles_fn):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.layout = TileLayout(w, h, tile_w, tile_h, self.world)
        self.device = device
        self.render_tiles_fn = render_tiles_fn
        self.mine = torch.zeros(self.layout.buffer_floats(), dtype=torch.float32, device=device)
        self.gathered = None
        self.image = None
        self.perm = None
        if self.rank == 0:
            if self.world > 1:
                # one block, rank-major: the gather writes into its slices, no concatenation afterwards
                self.gathered_flat = torch.empty(self.world * self.mine.numel(), dtype=torch.float32, device=device)
                self.gathered = list(self.gathered_flat.chunk(self.world))
            self.image = torch.zeros(h * w * 3, dtype=torch.float32, device=device)
            self.perm = torch.from_numpy(self.layout.scatter_index()).to(device)

    def step(self):
        """Render this rank's tiles, gather on rank 0, scatter to the image.  Returns the image view on
        rank 0 (h, w, 3) and None elsewhere.  Everything is enqueued on the current stream."""
        self.render_tiles_fn(self.mine, self.rank, self.world)
        if self.world > 1:
            dist.gather(self.mine, self.gathered, dst=0)
            if self.rank != 0:
                return None
            src = self.gathered_flat
        else:
            src = self.mine
        torch.index_select(src.view(-1, 3), 0, self.perm, out=self.image.view(-1, 3))
        return self.image.view(self.layout.h, self.layout.w, 3)


def device_render_fn(scene, cam, w, h, spp, seed, tile_w, tile_h, counters=None):
    """render_tiles_fn for a real GPU: rt_render_device on torch's current stream."""
    def fn(out, rank, world):
        stream = torch.cuda.current_stream(out.device)
        scene.render_device(cam, w, h, spp, seed, tile_w, tile_h, rank, world, out.data_ptr(),
                            counters.data_ptr() if counters is not None else 0, stream.cuda_stream)
    return fn


def new_counters(device):
    return torch.zeros(api.COUNTER_SLOTS, api.COUNTER_STRIDE, dtype=torch.int64, device=device)
