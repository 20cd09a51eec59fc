"""The N > 1 path on CPU: two (and three) gloo ranks run the tile sharding, gather and scatter of
raytracing_amd.parallel_render with a stand-in tile renderer that writes a known function of the pixel
coordinates; rank 0 must reassemble exactly that function.  The real renderer differs only in who
fills the tile buffer (rt_render_device on the rank's GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import api
from raytracing_amd.parallel_render import ProgressiveRenderer, ShardedRenderer, TileLayout


def expected_image(w, h):
    ys, xs = np.mgrid[0:h, 0:w]
    return np.stack([xs + 1000.0 * ys, xs * 0.5, ys * 0.25], -1).astype(np.float32)


def fake_render_fn(w, h, tw, th):
    """Fills the rank's tiles like rt_render does: tile-major, pixels outside the image 0."""
    def fn(out, rank, world):
        lay = TileLayout(w, h, tw, th, world)
        buf = np.zeros((lay.per_rank, th, tw, 3), np.float32)
        full = expected_image(w, h)
        for j in range(lay.local_tiles(rank)):
            t = rank + j * world
            x0, y0 = lay.tile_origin(t)
            blk = full[y0:y0 + th, x0:x0 + tw]
            buf[j, :blk.shape[0], :blk.shape[1]] = blk
        out.copy_(torch.from_numpy(buf.ravel()))
    return fn


def _worker(rank, world, port, w, h, tw, th, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for pipelined in (False, True):
            frame = {"i": 0}
            inner = fake_render_fn(w, h, tw, th)

            def fn(out, r_, wd):  # frame i is the known image + i, so a stale or mixed-up buffer shows
                inner(out, r_, wd)
                out += float(frame["i"])
                frame["i"] += 1
            r = ShardedRenderer(w, h, tw, th, torch.device("cpu"), fn, pipelined=pipelined)
            got = []
            for _ in range(5):  # five frames: both tile buffers are reused
                img = r.step()
                if img is not None:
                    got.append(img.numpy().copy())
            last = r.flush()
            if last is not None:
                got.append(last.numpy().copy())
            if rank == 0:
                ok = ok and len(got) == 5
                inside = expected_image(w, h)
                for i, g_ in enumerate(got):
                    ok = ok and bool(np.array_equal(g_, inside + np.float32(i)))
            else:
                ok = ok and not got
        if rank == 0:
            q.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,w,h,tw,th", [(2, 100, 37, 16, 8), (3, 64, 64, 16, 16), (2, 33, 9, 32, 8)])
def test_sharded_gather_reassembles_image(world, w, h, tw, th):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, tw, th, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def _progressive_worker(rank, world, port, w, h, tw, th, q):
    """Passes of 3 + 5 + 8 samples: a stand-in pass adds spp_pass times the known image to the sums and divides
    in the pass that reaches the total, like rt_render_pass_device; display() between passes must show the
    known image (sum / samples so far), and nothing but display() may talk to the other ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inner = fake_render_fn(w, h, tw, th)
        calls = []

        def pass_fn(out, r_, wd, sample0, spp_pass, spp_total):
            calls.append((sample0, spp_pass, spp_total))
            one = torch.zeros_like(out)
            inner(one, r_, wd)
            if sample0 == 0:
                out.zero_()
            out += one * float(spp_pass)
            if sample0 + spp_pass == spp_total:
                out /= float(spp_total)
        r = ProgressiveRenderer(w, h, tw, th, torch.device("cpu"), pass_fn, 16)
        ok = True
        want = expected_image(w, h)
        for k in (3, 5, 8):
            r.add_pass(k)
            img = r.display()
            if rank == 0:
                ok = ok and bool(np.allclose(img.numpy(), want, rtol=1e-6, atol=0))
            else:
                ok = ok and img is None
        ok = ok and calls == [(0, 3, 16), (3, 5, 16), (8, 8, 16)]
        try:
            r.add_pass(1)
            ok = False
        except ValueError:
            pass
        if rank == 0:
            q.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_progressive_passes_gather_only_on_display():
    world, w, h, tw, th = 2, 100, 37, 16, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_progressive_worker, args=(r, world, port, w, h, tw, th, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_layout_matches_c_library():
    for (w, h, tw, th, world) in [(1920, 1080, 16, 16, 8), (100, 37, 16, 8, 3), (33, 9, 32, 8, 2)]:
        lay = TileLayout(w, h, tw, th, world)
        assert lay.per_rank == api.tiles_per_rank(w, h, tw, th, 0, world)
        for r in range(world):
            assert lay.local_tiles(r) == api.tiles_per_rank(w, h, tw, th, r, world)
        # scatter index agrees with rt_tiles_to_image
        g = np.arange(world * lay.buffer_floats(), dtype=np.float32).reshape(world, -1)
        img = api.tiles_to_image(g, w, h, tw, th, world)
        idx = lay.scatter_index()
        assert np.array_equal(img.reshape(-1, 3), g.reshape(-1, 3)[idx])
