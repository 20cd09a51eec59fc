#!/usr/bin/env python3
"""rt_scene_refit against making the scene again, on the 10 M-triangle configuration (c5): the height field is bent,
the scene refitted (GPU time of the call, wall time with the upload) and rendered; the same geometry as a fresh scene
for comparison.  usage: bench_refit.py [--n 2237] [--amount 0.3]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from raytracing_amd import api, host, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2237)
ap.add_argument("--amount", type=float, default=0.3)
ap.add_argument("--spp", type=int, default=16)
a = ap.parse_args()
api.init(0)
d, keep = workloads.cornell_with_height_field(a.n)
pos, nrm = keep[2], keep[3]
t0 = time.perf_counter()
g = api.DeviceScene(d, keep=keep)
t1 = time.perf_counter()
info, tm = g.verify()
print(f"{d.n_triangles} triangles: rt_scene_create {1e3 * (t1 - t0):.1f} ms (device: sort {tm.device_sort_ms:.2f} + tree {tm.device_tree_ms:.2f} ms), {info.bvh_nodes} nodes")
cam = host.camera_default(1920, 1080)


def frame(scene, what):
    scene.render_tiles(cam, 1920, 1080, a.spp, 1, 16, 16, 0, 1)
    _, st = scene.render_tiles(cam, 1920, 1080, a.spp, 1, 16, 16, 0, 1)
    print(f"  {what}: {st.kernel_ms:.2f} ms per {a.spp}-spp frame, BVH nodes/ray {st.bvh_nodes / st.rays:.2f}, triangles/ray {st.bvh_tris / st.rays:.2f}")
    return st


frame(g, "as built")
x, z = pos[:, 0].astype(np.float64), pos[:, 2].astype(np.float64)
pos1 = pos.copy()
pos1[:, 1] = (-1.9 + (pos[:, 1].astype(np.float64) + 1.9) * (1.0 + a.amount) + 0.1 * a.amount * np.sin(4 * x) * np.cos(3 * z)).astype(np.float32)
t0 = time.perf_counter()
ms = g.refit(pos1, nrm)
t1 = time.perf_counter()
print(f"rt_scene_refit: {ms:.2f} ms on the GPU (with the upload of {pos1.nbytes * 2 / 1e6:.0f} MB), {1e3 * (t1 - t0):.1f} ms wall")
g.verify()
s_refit = frame(g, "refitted")
d.positions = pos1.ctypes.data_as(type(d.positions))
d.meshes[0].aabb_min[:] = pos1.min(0)
d.meshes[0].aabb_max[:] = pos1.max(0)
t0 = time.perf_counter()
f = api.DeviceScene(d, keep=keep + [pos1])
t1 = time.perf_counter()
print(f"the same geometry as a new scene: rt_scene_create {1e3 * (t1 - t0):.1f} ms")
s_fresh = frame(f, "fresh tree")
assert (s_refit.rays, s_refit.nodes) == (s_fresh.rays, s_fresh.nodes), "the two scenes must trace the same paths"
g.close(); f.close()
