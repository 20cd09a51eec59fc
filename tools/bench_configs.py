#!/usr/bin/env python3
"""The two large configurations, rehearsed on ONE GPU: the GPU renders the tile share of each of `world` ranks in
turn (what every rank of a world-GPU run does, minus the gather) and the slowest share projects the frame time.

  c4: the Cornell-with-mesh layout with a 1 002 528-triangle OFF mesh + the two Cornell spheres, 3840x2160, 256 spp
  c5: a 9 999 392-triangle height field + mirror and glass sphere + quad light in a box of squares, 1920x1080,
      progressive: passes of --pass-spp samples accumulated with rt_render_pass_device up to --spp

usage: bench_configs.py --config c4|c5 [--world 8] [--ranks 0,3,7] [--spp N] [--pass-spp 16]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracing_amd import api, host, parallel_render  # noqa: E402
from tests import scenegen  # noqa: E402
from tools import make_mesh_off  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", required=True, choices=["c4", "c5"])
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--ranks", default="")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--pass-spp", type=int, default=16)
ap.add_argument("--wave-times", action="store_true", help="needs RTAMD_LIB = a build with -DRT_WAVE_TIMES: clocks of the waves of one pass")
a = ap.parse_args()

api.init(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
if a.config == "c4":
    w, h, spp = 3840, 2160, a.spp or 256
    off = os.path.join(tempfile.gettempdir(), "grid_709.off")
    if not os.path.exists(off):
        make_mesh_off.write_off(off, *make_mesh_off.make(709))
    hs = host.HostScene.mesh_file(off, with_spheres=True)
    t1 = time.perf_counter()
    scene = hs.upload()
    pass_spp = spp
else:
    w, h, spp = 1920, 1080, a.spp or 64
    n = 2237
    rng = np.random.default_rng(1)
    xs = np.linspace(-1.9, 1.9, n, dtype=np.float32)
    gx, gz = np.meshgrid(xs, xs, indexing="xy")
    gy = (0.5 * np.exp(-1.5 * (gx * gx + gz * gz)) + 0.02 * rng.random((n, n), dtype=np.float32) - 1.9).astype(np.float32)
    pos = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange(n * n, dtype=np.uint32).reshape(n, n)
    p, q, r, s = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    tri = np.concatenate([np.stack([p, r, q], 1), np.stack([p, s, r], 1)]).astype(np.uint32)
    nrm = np.zeros_like(pos)
    nrm[:, 1] = 1.0
    box = host.HostScene.builtin(2)                      # Cornell walls, mirror + glass sphere, quad light
    d = api.copy_struct(api.SceneDesc, box.desc)
    mesh = (api.Mesh * 1)()
    mesh[0].first_vertex, mesh[0].n_vertices, mesh[0].first_triangle, mesh[0].n_triangles = 0, len(pos), 0, len(tri)
    mesh[0].material = 0
    mesh[0].aabb_min[:] = pos.min(0)
    mesh[0].aabb_max[:] = pos.max(0)
    d.n_meshes, d.meshes = 1, mesh
    d.n_vertices, d.n_triangles = len(pos), len(tri)
    import ctypes as C
    d.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
    d.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
    d.triangles = tri.ctypes.data_as(C.POINTER(C.c_uint32))
    t1 = time.perf_counter()
    scene = api.DeviceScene(d, keep=[box, mesh, pos, nrm, tri])
    pass_spp = a.pass_spp
t2 = time.perf_counter()
info, tm = scene.verify()
print(f"{a.config}: {info.bvh_nodes} BVH nodes (depth {info.bvh_max_depth}), {info.device_bytes / 1e6:.0f} MB of tables; host scene "
      f"{t1 - t0:.2f} s, rt_scene_create {t2 - t1:.3f} s (device build {tm.device_sort_ms + tm.device_tree_ms:.2f} ms)", flush=True)

cam = host.camera_default(w, h)
tw = th = 16
lay = parallel_render.TileLayout(w, h, tw, th, a.world)
out = torch.zeros(lay.buffer_floats(), dtype=torch.float32, device=dev)
cnt = parallel_render.new_counters(dev)
ranks = [int(x) for x in a.ranks.split(",")] if a.ranks else list(range(a.world))
stream = torch.cuda.current_stream(dev)
if a.wave_times:
    import ctypes as C
    L = api.lib()
    L.rt_debug_wave_times.argtypes = [C.c_void_p]
    per_wave = max(1, 64 // pass_spp) if pass_spp <= 32 else 1
    n_waves = lay.per_rank * ((tw * th + per_wave - 1) // per_wave)
    buf = torch.zeros(n_waves * 3, dtype=torch.int64, device=dev)
    assert L.rt_debug_wave_times(buf.data_ptr()) == 0
    for rank in ranks:
        for rep in range(2):
            buf.zero_()
            scene.render_pass_device(cam, w, h, min(pass_spp, spp), 0, spp, 1, tw, th, rank, a.world, out.data_ptr(), cnt.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(-1, 3)
        t = t[t[:, 1] > 0]
        t0 = t[:, 0].min()
        st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us at 100 MHz
        dur = en - st
        print(f"rank {rank}: {len(t)} waves, launch span {en.max():.0f} us, sum of wave times / 4096 slots {dur.sum() / 4096:.0f} us, "
              f"mean wave {dur.mean():.1f} us, max {dur.max():.0f} us; percentiles 50/90/99/99.9: {[round(float(np.percentile(dur, q))) for q in (50, 90, 99, 99.9)]}")
        for i in np.argsort(-en)[:10]:
            print(f"   ends {en[i]:.0f} us  started {st[i]:.0f} us  ran {dur[i]:.0f} us  pixel ({t[i, 2] & 0xffffffff}, {t[i, 2] >> 32})")
    scene.close()
    sys.exit(0)
worst = 0.0
for rank in ranks:
    cnt.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s0 in range(0, spp, pass_spp):
        scene.render_pass_device(cam, w, h, min(pass_spp, spp - s0), s0, spp, 1, tw, th, rank, a.world, out.data_ptr(), cnt.data_ptr(),
                                 stream.cuda_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    tot = cnt.sum(0).tolist()
    rays = tot[0]
    print(f"  rank {rank}/{a.world}: {ms:.1f} ms for {rays} rays = {rays / ms / 1e6:.2f} Grays/s on this GPU; BVH nodes/ray {tot[3] / rays:.2f}, "
          f"triangles/ray {tot[4] / rays:.2f}", flush=True)
    worst = max(worst, ms)
print(f"{a.config} {w}x{h} {spp} spp on {a.world} GPUs: slowest share {worst:.1f} ms per frame (gather not included)")
scene.close()
