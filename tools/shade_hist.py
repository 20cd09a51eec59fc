#!/usr/bin/env python3
"""How many lanes of a wave shade a path node together (debug build: make -C raytracing_amd/csrc OUT=variants_hist.so
EXTRA=-DRT_SHADE_HIST): histogram over all shade steps of one frame of a workload, and what sharing the 16 shadow rays of
a node among idle lanes could save.  usage: shade_hist.py [--config c2] [--world 8 --rank 3] [--spp N]"""
import argparse
import ctypes as C
import os
import sys

os.environ["RTAMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracing_amd", "csrc", "variants_hist.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from raytracing_amd import api, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--spp", type=int, default=0)
a = ap.parse_args()
api.init(0)
wl = workloads.make(a.config)
g = wl.upload()
L = api.lib()
L.rt_debug_shade_hist.argtypes = [C.c_void_p, C.c_int]
spp = a.spp or wl.pass_spp
g.render_tiles(wl.camera(), wl.w, wl.h, spp, 1, 16, 16, a.rank, a.world)
assert L.rt_debug_shade_hist(None, 1) == 0
_, st = g.render_tiles(wl.camera(), wl.w, wl.h, spp, 1, 16, 16, a.rank, a.world)
h = np.zeros(65, np.uint64)
assert L.rt_debug_shade_hist(h.ctypes.data_as(C.c_void_p), 0) == 0
h = h.astype(np.float64)
n = np.arange(65)
steps = h.sum()
iters = h[0]
h[0] = 0
steps = h.sum()
print(f"stats: {st.nodes} nodes, {st.samples} samples, {st.rays} rays, {st.rays_skipped} skipped; wave-level path-loop iterations {int(iters)}")
print(f"{a.config} {wl.w}x{wl.h} {spp} spp rank {a.rank}/{a.world}: {int(steps)} wave-level shade steps, {int((h * n).sum())} nodes shaded, kernel {st.kernel_ms:.2f} ms")
print("lanes shading together: share of the shade steps")
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 63), (64, 64)):
    print(f"  {lo:2d}..{hi:2d}: {h[lo:hi + 1].sum() / steps * 100:6.2f} %")
cost4 = np.where(n == 0, 0, np.where(n <= 16, 1, np.where(n <= 32, 2, 4)))   # batches of 4 rays per lane with 64 / n helpers (at most 4)
cost16 = np.where(n == 0, 0, np.where(n <= 4, 0.25, np.where(n <= 8, 0.5, cost4)))
print(f"shade steps cost 4 batches each today; with up to 4 lanes per node: {(h * cost4).sum() / (4 * steps) * 100:.1f} % of that, "
      f"with up to 16 lanes per node: {(h * cost16).sum() / (4 * steps) * 100:.1f} %")
g.close()
