"""rt_render_device_noalloc (the entry point that allocates nothing: camera and tile-order workspace in a static
device array) against rt_render_device, through the C-ABI: identical floats and counters on two calls in a row, the
second on a smaller image, then once more at the first size; sizes beyond the static array's capacity are refused."""
import numpy as np
import pytest

from raytracing_amd import api, host

pytestmark = pytest.mark.gpu


def _render(g, fn, cam, w, h, spp, tw, th, rank, world, seed):
    import torch
    dev = torch.device("cuda", 0)
    n = api.tiles_per_rank(w, h, tw, th, rank, world)
    out = torch.full((n * tw * th * 3,), -7.0, dtype=torch.float32, device=dev)   # stale contents must not matter
    cnt = torch.zeros(api.COUNTER_SLOTS, api.COUNTER_STRIDE, dtype=torch.int64, device=dev)
    fn(cam, w, h, spp, seed, tw, th, rank, world, out.data_ptr(), cnt.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy(), cnt.sum(0).cpu().numpy()


def test_noalloc_entry_point_equals_rt_render_device():
    api.init(0)
    g = host.HostScene.builtin(2).upload()       # mirror + glass sphere: the tile order workspace is in use
    try:
        for (w, h, spp, tw, th, rank, world) in ((320, 200, 70, 16, 16, 0, 1), (97, 41, 5, 16, 8, 1, 3), (320, 200, 70, 16, 16, 0, 1)):
            cam = host.camera_default(w, h)
            want, cw = _render(g, g.render_device, cam, w, h, spp, tw, th, rank, world, 5)
            got, cg = _render(g, g.render_device_noalloc, cam, w, h, spp, tw, th, rank, world, 5)
            assert want.view(np.uint32).tolist() == got.view(np.uint32).tolist(), (w, h)
            assert cw.tolist() == cg.tolist() and cw[0] > 0
            assert np.isfinite(got).all() and got.max() > 0.05
        # a share of more than RT_NOALLOC_MAX_TILES (2^17) tiles is refused before anything is launched
        import torch
        dummy = torch.zeros(16, dtype=torch.float32, device="cuda:0")
        with pytest.raises(api.RtError) as e:
            g.render_device_noalloc(host.camera_default(4096, 4096), 4096, 4096, 1, 1, 8, 8, 0, 1, dummy.data_ptr(), 0, 0)
        assert e.value.code == -5
    finally:
        g.close()
