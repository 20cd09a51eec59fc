#!/usr/bin/env python3
"""The reference's own configuration: 480 x 480, 50 samples per pixel, default camera, scenes 2 (Cornell box) and 3
(Cornell box with the star mesh, the scene of the committed rendu.ppm): kernel time and Mrays/s on one GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api, host
api.init(0)
for sid in (2, 3):
    g = host.HostScene.builtin(sid).upload()
    cam = host.camera_default(480, 480)
    for i in range(3):
        _, st = g.render_tiles(cam, 480, 480, 50, seed=1)
    print(f"scene {sid} 480x480 50 spp: {st.rays} rays, kernel {st.kernel_ms:.3f} ms, {st.rays / st.kernel_ms / 1e3:.1f} Mrays/s", flush=True)
