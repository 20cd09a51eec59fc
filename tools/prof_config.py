#!/usr/bin/env python3
"""One workload of raytracing_amd/workloads.py in a bare render loop, for rocprofv3 (--kernel-trace --stats, or one
--pmc pass per run): --steps frames of the configuration's own size (c5: four progressive passes of 16 spp per frame),
device buffers, nothing else on the GPU.  usage: prof_config.py --config c3|c4|c5|c2 [--steps 1] [--spp N] [--world 8 --rank 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracing_amd import api, parallel_render, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--pass-spp", type=int, default=0)
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--tile", type=int, default=16, help="tile edge (bench.py renders 8 x 8 tiles)")
ap.add_argument("--build", default="auto", help="auto | host | device: who builds the triangle BVH")
a = ap.parse_args()
api.init(0)
dev = torch.device("cuda", 0)
wl = workloads.make(a.config)
scene = wl.upload({"auto": None, "host": 0, "device": api.BUILD_BVH_DEVICE}[a.build])
w, h, spp = wl.w, wl.h, a.spp or wl.spp
pass_spp = min(a.pass_spp or wl.pass_spp, spp)
cam = wl.camera()
T = a.tile
lay = parallel_render.TileLayout(w, h, T, T, a.world)
out = torch.zeros(lay.buffer_floats(), dtype=torch.float32, device=dev)
cnt = parallel_render.new_counters(dev)
stream = torch.cuda.current_stream(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(a.steps):
    cnt.zero_()
    e0.record(stream)
    for s0 in range(0, spp, pass_spp):
        scene.render_pass_device(cam, w, h, min(pass_spp, spp - s0), s0, spp, 1, T, T, a.rank, a.world, out.data_ptr(), cnt.data_ptr(), stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    t = cnt.sum(0).tolist()
    ms = e0.elapsed_time(e1)
    traced = t[0] - t[5]
    alg = t[3] * 64 + t[4] * 48 + traced * 32 + t[1] * 12
    print(f"step {i}: {a.config} {w}x{h} {spp} spp (passes of {pass_spp}), rank {a.rank}/{a.world}: {traced} rays traced ({t[0]} reference rays), "
          f"{ms:.3f} ms, {traced / ms / 1e6:.3f} Grays/s; BVH nodes/ray {t[3] / traced:.2f}, triangles/ray {t[4] / traced:.2f}, "
          f"algorithmic {alg / traced:.0f} B/ray = {alg / ms / 1e6:.1f} GB/s", flush=True)
scene.close()
