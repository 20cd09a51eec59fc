#!/usr/bin/env python3
"""Kernel time of the batch Scene::rayTrace entry point (rt_trace, k_trace) on one frame's worth of primary rays: the
Cornell box (kernel variant 0) and the many-spheres scene (variant 1).  usage: bench_trace.py [--w 960 --h 540 --spp 4]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api, host  # noqa: E402
from tests import scenegen  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=960)
ap.add_argument("--h", type=int, default=540)
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--only", default="", help="cornell | kd: one of the two scenes")
a = ap.parse_args()
api.init(0)
w, h, spp = a.w, a.h, a.spp
rng = np.random.default_rng(1)
ys, xs = np.mgrid[0:h, 0:w]
pix = np.repeat((xs + ys * w).ravel().astype(np.uint32), spp)
smp = np.tile(np.arange(spp, dtype=np.uint32), w * h)
u = ((np.repeat(xs.ravel(), spp) + rng.random(pix.size)) / w).astype(np.float32)
v = ((np.repeat(ys.ravel(), spp) + rng.random(pix.size)) / h).astype(np.float32)
org, d = api.generate_rays(host.camera_default(w, h), u, v)
d = api.normalize_dirs(d)
desc, keep = scenegen.random_objects(api, n_spheres=60, n_squares=30, seed=7)
for name, make in (("cornell", lambda: host.HostScene.builtin(2).upload()), ("kd 60+30", lambda: api.DeviceScene(desc, keep=keep))):
    if a.only and not name.startswith(a.only):
        continue
    g = make()
    best = None
    for i in range(5):
        _, st = g.trace(org, d, pix, smp)
        best = st if best is None or st.kernel_ms < best.kernel_ms else best
    st = best
    print(f"{name}: {len(org)} paths, {st.rays} rays, k_trace {st.kernel_ms:.3f} ms, {st.rays / st.kernel_ms / 1e6:.2f} Grays/s", flush=True)
    g.close()
