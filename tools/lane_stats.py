#!/usr/bin/env python3
"""How many lanes of a wave run each marked region of the render kernel (debug build: make -C raytracing_amd/csrc
OUT=variants_lanes.so EXTRA=-DRT_LANE_STATS): visits by a wave and lanes per visit, over one frame of a workload.
usage: lane_stats.py [--config c2] [--spp N]"""
import argparse
import ctypes as C
import os
import sys

os.environ["RTAMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracing_amd", "csrc", "variants_lanes.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from raytracing_amd import api, workloads  # noqa: E402

NAMES = {0: "shade: a path node", 1: "a batch of light samples", 2: "sample rules of one square", 3: "a sphere: discriminants of a batch",
         4: "a sphere: certificates", 5: "a sphere: roots of one ray", 6: "a square: cheap test of a batch", 7: "a square: exact test of one ray",
         8: "light term of one ray", 9: "closest hit of one ray", 10: "rays a mirror / glass hit sends on", 11: "glass node not shaded"}
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--spp", type=int, default=0)
a = ap.parse_args()
api.init(0)
wl = workloads.make(a.config)
g = wl.upload()
L = api.lib()
L.rt_debug_lane_stats.argtypes = [C.c_void_p, C.c_int]
spp = a.spp or wl.pass_spp
g.render_tiles(wl.camera(), wl.w, wl.h, spp, 1, 8, 8, 0, 1)
assert L.rt_debug_lane_stats(None, 1) == 0
_, st = g.render_tiles(wl.camera(), wl.w, wl.h, spp, 1, 8, 8, 0, 1)
h = np.zeros(64, np.uint64)
assert L.rt_debug_lane_stats(h.ctypes.data_as(C.c_void_p), 0) == 0
h = h.astype(np.float64)
waves = wl.w * wl.h * spp / 64
print(f"{a.config} {wl.w}x{wl.h} {spp} spp: {st.rays} rays, {st.nodes} nodes, kernel {st.kernel_ms:.2f} ms (debug build)")
print(f"{'region':40s} {'visits / 64 samples':>20s} {'lanes / visit':>14s}")
for k, name in NAMES.items():
    if h[2 * k]:
        print(f"{name:40s} {h[2 * k] / waves:20.2f} {h[2 * k + 1] / h[2 * k]:14.1f}")
g.close()
