#!/usr/bin/env python3
"""Secondary benchmark: a scene of many small objects (60 spheres + 30 squares), for which the object kd-tree has
internal nodes and the kd render variant runs.   usage: bench_kd.py [--width 960 --height 540 --spp 64] [--mesh N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api, host  # noqa: E402
from tests import scenegen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=960)
ap.add_argument("--height", type=int, default=540)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--diffuse", action="store_true", help="no mirror / glass spheres: every path has one node")
ap.add_argument("--mesh", type=int, default=0, help="add an n x n height-field mesh (BVH): kd + BVH variant")
ap.add_argument("--leaf", type=int, default=30, help="kd_max_leaf (the reference uses 30; 1000 keeps the root a leaf: brute force over all objects)")
a = ap.parse_args()
mesh = scenegen.grid_mesh(a.mesh, seed=2) if a.mesh else None
desc, keep = scenegen.random_objects(api, n_spheres=60, n_squares=30, seed=7, with_mesh=mesh, kd_max_leaf=a.leaf, diffuse_only=a.diffuse)
info = api.scene_check(desc)
print(f"kd nodes {info.kd_nodes}, depth {info.kd_max_depth}, kernel variant {info.kernel_variant}", flush=True)
api.init(0)
g = api.DeviceScene(desc, keep=keep)
cam = host.camera_default(a.width, a.height)
for i in range(a.steps):
    _, st = g.render_tiles(cam, a.width, a.height, a.spp, seed=1)
    print(f"step {i}: {st.rays} rays ({st.rays / st.samples:.2f}/sample), kernel {st.kernel_ms:.3f} ms, {st.rays / st.kernel_ms / 1e3:.1f} Mrays/s", flush=True)
g.close()
