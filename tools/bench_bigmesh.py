#!/usr/bin/env python3
"""Scaling check of the triangle-BVH path: an n x n height field (2 (n-1)^2 triangles, built in memory, no OFF
file) inside a box of squares with two spheres and the quad light, rendered at 1080p.  Prints build time,
table size, Mrays/s and BVH work per ray.   usage: bench_bigmesh.py [--n 2237] [--spp 16] [--steps 2] [--build host|device|both]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api, host  # noqa: E402
from tests import scenegen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2237)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--build", default="host", choices=["host", "device", "both"])
a = ap.parse_args()

n = a.n
rng = np.random.default_rng(1)
xs = np.linspace(-1.2, 1.2, n, dtype=np.float32)
gx, gz = np.meshgrid(xs, xs, indexing="xy")
gy = (0.5 * np.exp(-1.5 * (gx * gx + gz * gz)) + 0.02 * rng.random((n, n), dtype=np.float32) - 1.6).astype(np.float32)
pos = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(np.float32)
idx = np.arange(n * n, dtype=np.uint32).reshape(n, n)
p, q, r, s = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
tri = np.concatenate([np.stack([p, r, q], 1), np.stack([p, s, r], 1)]).astype(np.uint32)
nrm = np.zeros_like(pos)
nrm[:, 1] = 1.0
desc, keep = scenegen.random_objects(api, n_spheres=2, n_squares=6, seed=11, with_mesh=(pos, nrm, tri))
print(f"mesh: {len(tri)} triangles, {len(pos)} vertices", flush=True)
api.init(0)
cam = host.camera_default(a.width, a.height)
for build in (["host", "device"] if a.build == "both" else [a.build]):
    t0 = time.perf_counter()
    g = api.DeviceScene(desc, keep=keep, flags=api.BUILD_BVH_DEVICE if build == "device" else 0)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    info, tm = g.verify()   # reads the tables back and checks them on the host
    print(f"[{build}] rt_scene_create {dt:.3f} s (host part {tm.host_ms:.0f} ms; device: keys + sort + permutation {tm.device_sort_ms:.2f} ms, "
          f"hierarchy + boxes {tm.device_tree_ms:.2f} ms); verified in {time.perf_counter() - t0:.1f} s: {info.bvh_nodes} BVH nodes, "
          f"depth {info.bvh_max_depth}, {info.device_bytes / 1e6:.0f} MB of tables", flush=True)
    for i in range(a.steps):
        _, st = g.render_tiles(cam, a.width, a.height, a.spp, seed=1)
        print(f"[{build}] step {i}: {st.rays} rays, kernel {st.kernel_ms:.2f} ms, {st.rays / st.kernel_ms / 1e3:.1f} Mrays/s; BVH nodes/ray "
              f"{st.bvh_nodes / st.rays:.2f}, triangles/ray {st.bvh_tris / st.rays:.2f}", flush=True)
    g.close()
