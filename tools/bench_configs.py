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
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracing_amd import api, host, parallel_render, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", required=True, choices=["c2", "c3", "c4", "c5"])
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--ranks", default="")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--pass-spp", type=int, default=16)
ap.add_argument("--wave-times", action="store_true", help="needs RTAMD_LIB = a build with -DRT_WAVE_TIMES: clocks of the waves of one pass")
a = ap.parse_args()

api.init(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
wl = workloads.make(a.config)
w, h, spp = wl.w, wl.h, a.spp or wl.spp
t1 = time.perf_counter()
scene = wl.upload()
pass_spp = spp if a.config == "c4" else a.pass_spp
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
    n_waves = lay.per_rank * ((tw * th + per_wave - 1) // per_wave) + 4096   # + the worker waves, recorded behind the pixel waves
    buf = torch.zeros(n_waves * 3, dtype=torch.int64, device=dev)
    assert L.rt_debug_wave_times(buf.data_ptr()) == 0
    items = torch.zeros(3 << 20, dtype=torch.int64, device=dev)   # one record per worker batch (by first sample / 16)
    L.rt_debug_worker_items.argtypes = [C.c_void_p]
    assert L.rt_debug_worker_items(items.data_ptr()) == 0
    for rank in ranks:
        for rep in range(2):
            buf.zero_()
            items.zero_()
            scene.render_pass_device(cam, w, h, min(pass_spp, spp), 0, spp, 1, tw, th, rank, a.world, out.data_ptr(), cnt.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(-1, 3)
        t = t[t[:, 1] > 0]
        t0 = t[:, 0].min()
        wk = t[((t[:, 2] >> 44) & 0xfffff) == 0xfffff]
        if len(wk):
            busy = wk[((wk[:, 2] >> 24) & 0xfffff) > 0]
            print(f"rank {rank}: {len(wk)} worker waves, {len(busy)} traced samples ({int(((busy[:, 2] >> 24) & 0xfffff).sum())} in all); they end at "
                  f"{[round(float(x)) for x in np.percentile((busy[:, 1] - t0) / 100.0, (50, 90, 100))] if len(busy) else []} us (50 / 90 / 100 %)")
        it = items.cpu().numpy().reshape(-1, 3)
        it = it[it[:, 1] > 0]
        if len(it):
            ist, idur = (it[:, 0] - t0) / 100.0, (it[:, 1] - it[:, 0]) / 100.0
            nodes, shaded, ns = it[:, 2] & 0xffff, (it[:, 2] >> 16) & 0xffff, (it[:, 2] >> 32) & 0xff
            print(f"rank {rank}: {len(it)} worker batches of {np.bincount(ns).nonzero()[0].tolist()} samples; us per batch 50/90/99/max "
                  f"{[round(float(x)) for x in np.percentile(idur, (50, 90, 99, 100))]}; nodes per batch 50/90/99/max {[int(x) for x in np.percentile(nodes, (50, 90, 99, 100))]}; "
                  f"shaded {[int(x) for x in np.percentile(shaded, (50, 90, 99, 100))]}; us per node {idur.sum() / nodes.sum():.2f}; corr(us, nodes) {np.corrcoef(idur, nodes)[0, 1]:.2f}")
            iexp = ((it[:, 2] >> 40) & 0xffffff) / 100.0   # us until the tree was expanded (closest hits); the rest is shading
            print(f"   expansion share of the batch time: all batches {iexp.sum() / idur.sum() * 100:.0f} %, the slowest tenth {iexp[idur > np.percentile(idur, 90)].sum() / idur[idur > np.percentile(idur, 90)].sum() * 100:.0f} %")
            for i in np.argsort(-(ist + idur))[:8]:
                print(f"   batch ends {ist[i] + idur[i]:.0f} us  started {ist[i]:.0f}  ran {idur[i]:.0f} us (expansion {iexp[i]:.0f})  {nodes[i]} nodes, {shaded[i]} shaded, {ns[i]} samples")
            late = ist > np.percentile(ist, 90)
            print(f"   the last tenth of the batches to start: start {ist[late].min():.0f} .. {ist[late].max():.0f} us, us per batch 50/max {np.percentile(idur[late], 50):.0f} / {idur[late].max():.0f}")
        t = t[((t[:, 2] >> 44) & 0xfffff) != 0xfffff]
        st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us at 100 MHz
        dur = en - st
        wave_bvh = (t[:, 2] >> 44) & 0xfffff
        print(f"rank {rank}: waves that fetched fewer than 64 BVH nodes: {int((wave_bvh == 0).sum())} of {len(t)} ({(wave_bvh == 0).mean() * 100:.1f} %), "
              f"their share of the summed wave time {dur[wave_bvh == 0].sum() / dur.sum() * 100:.1f} %")
        print(f"rank {rank}: {len(t)} waves, launch span {en.max():.0f} us, sum of wave times / 4096 slots {dur.sum() / 4096:.0f} us, "
              f"mean wave {dur.mean():.1f} us, max {dur.max():.0f} us; percentiles 50/90/99/99.9: {[round(float(np.percentile(dur, q))) for q in (50, 90, 99, 99.9)]}")
        for i in np.argsort(-en)[:10]:
            print(f"   ends {en[i]:.0f} us  started {st[i]:.0f} us  ran {dur[i]:.0f} us  pixel ({t[i, 2] & 0xfff}, {(t[i, 2] >> 12) & 0xfff}) path nodes {(t[i, 2] >> 24) & 0xfffff} BVH nodes {((t[i, 2] >> 44) & 0xfffff) * 64}")
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
