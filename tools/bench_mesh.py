#!/usr/bin/env python3
"""Secondary benchmark: the Cornell-with-mesh layout (setup_cornell_box_meshes) with a synthetic OFF mesh of
~100k triangles loaded through the host Mesh class, rendered on one GPU.  Prints Mrays/s and the BVH size.
usage: bench_mesh.py [--n 225] [--width 1920 --height 1080 --spp 8] [--spheres] [--steps 3] [--build host|device]"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import api, host  # noqa: E402
from tools import make_mesh_off  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=225)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--spheres", action="store_true")
ap.add_argument("--build", default="host", choices=["host", "device"])
a = ap.parse_args()
off = os.path.join(tempfile.gettempdir(), f"grid_{a.n}.off")
if not os.path.exists(off):
    p, t = make_mesh_off.make(a.n)
    make_mesh_off.write_off(off, p, t)
import time  # noqa: E402
api.init(0)
t0 = time.perf_counter()
hs = host.HostScene.mesh_file(off, with_spheres=a.spheres)
t1 = time.perf_counter()
print(f"mesh: {hs.desc.n_triangles} triangles, {hs.desc.n_vertices} vertices; loadOFF + normals + flatten {t1 - t0:.2f} s", flush=True)
g = hs.upload(flags=api.BUILD_BVH_DEVICE if a.build == "device" else 0)
dt = time.perf_counter() - t1
info, tm = g.verify()
print(f"rt_scene_create ({a.build} BVH build + upload) {dt:.3f} s (host part {tm.host_ms:.0f} ms, device {tm.device_sort_ms + tm.device_tree_ms:.2f} ms); "
      f"{info.bvh_nodes} nodes, depth {info.bvh_max_depth}", flush=True)
cam = host.camera_default(a.width, a.height)
for i in range(a.steps):
    _, st = g.render_tiles(cam, a.width, a.height, a.spp, seed=1)
    bytes_per_ray = (st.bvh_nodes * 64 + st.bvh_tris * 48) / st.rays + 32 + 12 * st.samples / st.rays
    print(f"step {i}: {st.rays} rays ({st.rays / st.samples:.2f}/sample), kernel {st.kernel_ms:.3f} ms, {st.rays / st.kernel_ms / 1e3:.1f} Mrays/s; "
          f"BVH nodes/ray {st.bvh_nodes / st.rays:.2f}, triangles/ray {st.bvh_tris / st.rays:.2f}, algorithmic {bytes_per_ray:.0f} B/ray "
          f"= {bytes_per_ray * st.rays / st.kernel_ms / 1e9:.2f} TB/s", flush=True)
g.close()
