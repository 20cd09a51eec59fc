#!/usr/bin/env python3
"""Rehearsal of the N > 1 exchange step with the real backend (nccl = RCCL): every rank fills its tile buffer with
a known function of the pixel position on the GPU, ShardedRenderer gathers (pipelined) and scatters, rank 0
checks the image.  Launch: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1
tools/nccl_selftest.py [--same-device]   (--same-device puts every rank on GPU 0: only a smoke test, RCCL may
refuse two ranks on one GPU)."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.parallel_render import ShardedRenderer, TileLayout  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--same-device", action="store_true")
a = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
w, h, tw, th = 200, 90, 16, 16
ys, xs = np.mgrid[0:h, 0:w]
want = np.stack([xs + 1000.0 * ys, xs * 0.5, ys * 0.25], -1).astype(np.float32)
frame = {"i": 0}


def fill(out, r, wd):
    lay = TileLayout(w, h, tw, th, wd)
    buf = np.zeros((lay.per_rank, th, tw, 3), np.float32)
    for j in range(lay.local_tiles(r)):
        x0, y0 = lay.tile_origin(r + j * wd)
        blk = want[y0:y0 + th, x0:x0 + tw]
        buf[j, :blk.shape[0], :blk.shape[1]] = blk
    out.copy_(torch.from_numpy(buf.ravel()).to(out.device) + float(frame["i"]))
    frame["i"] += 1


ren = ShardedRenderer(w, h, tw, th, dev, fill)
got = []
for _ in range(4):
    img = ren.step()
    if img is not None:
        got.append(img.cpu().numpy().copy())
last = ren.flush()
if last is not None:
    got.append(last.cpu().numpy().copy())
torch.cuda.synchronize()
ok = True
if rank == 0:
    ok = len(got) == 4 and all(np.array_equal(g, want + np.float32(i)) for i, g in enumerate(got))
    print("nccl selftest", "OK" if ok else "FAILED", f"({world} ranks)", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
