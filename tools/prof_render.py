#!/usr/bin/env python3
"""Minimal render loop for profiling (no torch): Cornell 1080p at --spp, --steps launches of k_render
through rt_render.  Used under rocprofv3 (--kernel-trace --stats, or one --pmc pass per run)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=2)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--world", type=int, default=1, help="render only the tiles of rank 0 of this many ranks")
a = ap.parse_args()
api.init(0)
g = host.HostScene.builtin(a.scene).upload()
cam = host.camera_default(a.width, a.height)
for i in range(a.steps):
    _, st = g.render_tiles(cam, a.width, a.height, a.spp, seed=1, rank=0, world=a.world)
    print(f"step {i}: {st.rays} rays, kernel {st.kernel_ms:.3f} ms, {st.rays / st.kernel_ms / 1e6:.3f} Grays/s", flush=True)
g.close()
