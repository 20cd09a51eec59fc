#!/usr/bin/env python3
"""Wall time per bench step when this GPU renders only 1/world of the tiles (what one rank of an N-GPU run
does), without the gather, for every rank: shows the fixed per-step cost and the load balance of the tile split."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracing_amd import api, host, parallel_render  # noqa: E402

api.init(0)
dev = torch.device("cuda", 0)
scene = host.HostScene.builtin(2).upload()
tw = th = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w, h, spp = 1920, 1080, 64
cam = host.camera_default(w, h)
for world in (1, 2, 4, 8, 8, 8):
    lay = parallel_render.TileLayout(w, h, tw, th, world)
    out = torch.zeros(lay.buffer_floats(), dtype=torch.float32, device=dev)
    cnt = parallel_render.new_counters(dev)
    fn = parallel_render.device_render_fn(scene, cam, w, h, spp, 1, tw, th, cnt)
    per_rank = []
    for rank in range(world):
        for _ in range(2):
            fn(out, rank, world)
        torch.cuda.synchronize()
        n = 10
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(n):
            fn(out, rank, world)
            evs[i + 1].record()
        torch.cuda.synchronize()
        per_rank.append((time.perf_counter() - t0) / n * 1e3)
        if world == 8:
            print(f"   rank {rank}: per launch ms", [round(evs[i].elapsed_time(evs[i + 1]), 2) for i in range(n)])
    print(f"world {world}: per-rank ms {[round(x, 3) for x in per_rank]}  max x world = {max(per_rank) * world:.2f} ms, "
          f"sum = {sum(per_rank):.2f} ms")
