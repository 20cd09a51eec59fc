#!/usr/bin/env python3
"""The deferred shadow-ray walks of one frame of a mesh workload (debug build: make -C raytracing_amd/csrc
OUT=variants_walk.so EXTRA=-DRT_WALK_STATS): calls, rays per call, the longest walk of a call against the mean.
usage: walk_stats.py [--config c3] [--world 1 --rank 0] [--spp N]"""
import argparse
import ctypes as C
import os
import sys

os.environ["RTAMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracing_amd", "csrc", "variants_walk.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from raytracing_amd import api, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--spp", type=int, default=0)
a = ap.parse_args()
api.init(0)
wl = workloads.make(a.config)
g = wl.upload()
L = api.lib()
L.rt_debug_walk_stats.argtypes = [C.c_void_p, C.c_int]
spp = a.spp or wl.pass_spp
g.render_tiles(wl.camera(), wl.w, wl.h, spp, 1, 16, 16, a.rank, a.world)
assert L.rt_debug_walk_stats(None, 1) == 0
_, st = g.render_tiles(wl.camera(), wl.w, wl.h, spp, 1, 16, 16, a.rank, a.world)
h = np.zeros(128, np.uint64)
assert L.rt_debug_walk_stats(h.ctypes.data_as(C.c_void_p), 0) == 0
h = h.astype(np.float64)
calls, rays, mx, sm, lanes = h[0], h[1], h[2], h[3], h[4]
print(f"{a.config} {wl.w}x{wl.h} {spp} spp rank {a.rank}/{a.world}: kernel {st.kernel_ms:.2f} ms, {st.nodes} path nodes shaded, {st.rays} rays")
print(f"{int(calls)} walk calls ({calls / (st.nodes / 64):.2f} per 64 shaded nodes), {rays / calls:.1f} rays per call on {lanes / calls:.1f} lanes; "
      f"node units per ray {sm / rays:.1f}; longest walk of a call {mx / calls:.1f} units = {mx / calls / (sm / rays):.1f} x the mean ray; "
      f"lane use during the walks {sm / (mx * 64) * 100:.1f} %")
print(f"shader cycles inside the walk calls: {h[5]:.3e} = {h[5] / calls:.0f} per call ({h[5] / 2.4e9 / 4096 * 1e3:.2f} ms of all 4096 wave slots at 2.4 GHz); "
      f"pool loop: {h[6] / calls:.1f} iterations and {h[7] / calls:.1f} fetch phases per call, {h[5] / max(h[6], 1):.0f} cycles per iteration")
for k, name in enumerate(("fetch a ray", "visit a wide node", "test a leaf")):
    if h[114 + k]:
        print(f"  trips that {name:18s}: {h[114 + k] / calls:6.1f} per call, {h[117 + k] / h[114 + k]:5.1f} lanes served, {h[120 + k] / h[114 + k]:6.0f} cycles each")
if h[124]:
    print(f"  a wide node's fetch, issue to arrival (first lane of the wave, {int(h[124])} fetches): {h[123] / h[124]:.0f} cycles")
print("rays per call: share of calls")
for b in range(40):
    if h[8 + b]:
        print(f"  {8 * b:3d}..{8 * b + 7:3d}{'+' if b == 39 else ' '}: {h[8 + b] / calls * 100:6.2f} %")
print("longest walk of a call (64-byte node units): share of calls")
for b in range(40):
    if h[48 + b]:
        print(f"  {16 * b:3d}..{16 * b + 15:3d}{'+' if b == 39 else ' '}: {h[48 + b] / calls * 100:6.2f} %")
g.close()
