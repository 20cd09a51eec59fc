"""The benchmark workloads: the five configurations the project's baseline names, as host scenes ready to upload.

  c1  built-in Cornell box (scene 2), the reference's own 480 x 480, 50 spp
  c2  the same box at 1920 x 1080, 64 spp                                          (headline)
  c3  Cornell-with-mesh layout, OFF mesh of 100 352 triangles + the two spheres, 1920 x 1080, 64 spp
  c4  the same layout with 1 002 528 triangles, 3840 x 2160, 256 spp
  c5  9 999 392-triangle height field on the floor of the Cornell box with the mirror and the glass sphere and the
      quad light, 1920 x 1080, 64 spp accumulated in progressive passes of 16

Scene set-up runs through the product's C++ host classes (host.HostScene); only the synthetic geometry is made here
(numpy, seeded).  No GPU call and no torch import in this module."""
import ctypes as C
import os
import tempfile

import numpy as np

from . import api, host

SPECS = {
    "c1": dict(w=480, h=480, spp=50, pass_spp=50, what="cornell_box(scene 2)"),
    "c2": dict(w=1920, h=1080, spp=64, pass_spp=64, what="cornell_box(scene 2)"),
    "c3": dict(w=1920, h=1080, spp=64, pass_spp=64, what="cornell box + OFF mesh of 100352 triangles + mirror and glass sphere"),
    "c4": dict(w=3840, h=2160, spp=256, pass_spp=256, what="cornell box + OFF mesh of 1002528 triangles + mirror and glass sphere"),
    "c5": dict(w=1920, h=1080, spp=64, pass_spp=16,
               what="cornell box + height field of 9999392 triangles + mirror and glass sphere, progressive passes of 16 spp"),
}


def dome_mesh(n, seed=1):
    """A bumpy n x n vertex height field folded into a dome, 2 (n-1)^2 triangles (n = 225: 100 352, n = 709: 1 002 528).
    Returns float64 positions and int triangle indices; the OFF writer rounds them to 7 significant digits."""
    rng = np.random.default_rng(seed)
    u = np.linspace(-1.0, 1.0, n)
    gx, gz = np.meshgrid(u, u, indexing="xy")
    r2 = gx * gx + gz * gz
    gy = 0.6 * np.exp(-1.5 * r2) + 0.04 * rng.random((n, n)) + 0.05 * np.sin(9 * gx) * np.cos(7 * gz)
    pos = np.stack([gx, gy, gz], -1).reshape(-1, 3)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    tri = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)])
    return pos, tri


def write_off(path, pos, tri):
    """The format Mesh::loadOFF reads (R/src/Mesh.cpp:21-39): header, counts, vertices, '3 i j k' faces."""
    tmp = path + f".tmp{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(f"OFF\n{len(pos)} {len(tri)} 0\n\n")
        np.savetxt(f, pos, fmt="%.7g")
        np.savetxt(f, np.concatenate([np.full((len(tri), 1), 3), tri], 1), fmt="%d")
    os.replace(tmp, path)


def dome_off(n, seed=1, directory=None):
    """Path of the OFF file of dome_mesh(n, seed), written on first use."""
    path = os.path.join(directory or tempfile.gettempdir(), f"rtamd_dome_{n}_{seed}.off")
    if not os.path.exists(path):
        write_off(path, *dome_mesh(n, seed))
    return path


def height_field(n=2237, seed=1):
    """c5's mesh: n x n vertices over [-1.9, 1.9]^2 just above the floor (y = -2) of the Cornell box, a smooth hill plus
    seeded roughness; 2 (n-1)^2 triangles (n = 2237: 9 999 392).  float32 positions, +y normals, uint32 triangles."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-1.9, 1.9, n, dtype=np.float32)
    gx, gz = np.meshgrid(xs, xs, indexing="xy")
    gy = (0.5 * np.exp(-1.5 * (gx * gx + gz * gz)) + 0.02 * rng.random((n, n), dtype=np.float32) - 1.9).astype(np.float32)
    pos = np.ascontiguousarray(np.stack([gx, gy, gz], -1).reshape(-1, 3), np.float32)
    idx = np.arange(n * n, dtype=np.uint32).reshape(n, n)
    p, q, r, s = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    tri = np.ascontiguousarray(np.concatenate([np.stack([p, r, q], 1), np.stack([p, s, r], 1)]), np.uint32)
    nrm = np.zeros_like(pos)
    nrm[:, 1] = 1.0
    return pos, nrm, tri


class Workload:
    """desc: the rt_scene_desc (api.SceneDesc) of the scene; keep: what its pointers refer to."""

    def __init__(self, name, desc, keep, w, h, spp, pass_spp, what):
        self.name, self.desc, self.keep = name, desc, keep
        self.w, self.h, self.spp, self.pass_spp, self.what = w, h, spp, pass_spp, what

    def upload(self, flags=None):
        return api.DeviceScene(self.desc, keep=self, flags=flags)

    def camera(self, w=None, h=None):
        return host.camera_default(w or self.w, h or self.h)


def cornell_with_height_field(n=2237, seed=1):
    """Scene of c5 (any n): the flattened built-in Cornell box with one mesh appended."""
    pos, nrm, tri = height_field(n, seed)
    box = host.HostScene.builtin(2)
    d = api.copy_struct(api.SceneDesc, box.desc)
    mesh = (api.Mesh * 1)()
    mesh[0].first_vertex, mesh[0].n_vertices, mesh[0].first_triangle, mesh[0].n_triangles = 0, len(pos), 0, len(tri)
    mesh[0].material = 0
    mesh[0].aabb_min[:] = pos.min(0)
    mesh[0].aabb_max[:] = pos.max(0)
    d.n_meshes, d.meshes = 1, mesh
    d.n_vertices, d.n_triangles = len(pos), len(tri)
    d.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
    d.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
    d.triangles = tri.ctypes.data_as(C.POINTER(C.c_uint32))
    return d, [box, mesh, pos, nrm, tri]


def make(name, mesh_n=None):
    """Workload `name` (c1 .. c5).  mesh_n overrides the grid size of the synthetic mesh (tests use small ones)."""
    sp = SPECS[name]
    if name in ("c1", "c2"):
        hs = host.HostScene.builtin(2)
        desc, keep = hs.desc, [hs]
    elif name in ("c3", "c4"):
        hs = host.HostScene.mesh_file(dome_off(mesh_n or (225 if name == "c3" else 709)), with_spheres=True)
        desc, keep = hs.desc, [hs]
    else:
        desc, keep = cornell_with_height_field(mesh_n or 2237)
    return Workload(name, desc, keep, sp["w"], sp["h"], sp["spp"], sp["pass_spp"], sp["what"])
