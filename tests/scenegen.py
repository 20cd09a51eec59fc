"""Seeded synthetic scenes as rt_scene_desc (ctypes, binding-agnostic): many small objects so that
the object kd-tree has internal nodes, and synthetic triangle meshes."""
import ctypes as C

import numpy as np


def random_objects(api_mod, n_spheres=40, n_squares=24, seed=3, with_mesh=None, point_light=False, kd_max_leaf=30,
                   kd_max_depth=15, second_mesh=None, diffuse_only=False):
    """A box-like world of small spheres and squares; api_mod is oracle.orc or raytracing_amd.api (same layouts).
    Returns (desc, keepalive)."""
    rng = np.random.default_rng(seed)
    mats = (api_mod.Material * (n_spheres + n_squares + (1 if with_mesh else 0)))()
    for i, m in enumerate(mats):
        m.diffuse[:] = rng.uniform(0.1, 1.0, 3).astype(np.float32)
        m.specular[:] = rng.uniform(0.0, 1.0, 3).astype(np.float32)
        m.shininess = float(rng.choice([0.0, 16.0, 20.0, 7.5]))
        m.index_medium = float(rng.choice([1.0, 1.3, 1.5]))
        m.type = int(rng.choice([0, 0, 1, 2])) if i < n_spheres else 0
        if diffuse_only:
            m.type = 0
    sph = (api_mod.Sphere * max(n_spheres, 1))()
    for i in range(n_spheres):
        sph[i].center[:] = rng.uniform(-1.8, 1.8, 3).astype(np.float32)
        sph[i].radius = float(np.float32(rng.uniform(0.05, 0.3)))
        sph[i].material = i
    sq = (api_mod.Square * max(n_squares, 1))()
    for i in range(n_squares):
        c = rng.uniform(-1.9, 1.9, 3)
        a = rng.normal(size=3)
        a /= np.linalg.norm(a)
        b = np.cross(a, rng.normal(size=3))
        b /= np.linalg.norm(b)
        w, h = rng.uniform(0.2, 1.0, 2)
        v0 = c - 0.5 * w * a - 0.5 * h * b
        pts = [v0, v0 + w * a, v0 + w * a + h * b, v0 + h * b]
        for k in range(4):
            sq[i].v[k][:] = np.asarray(pts[k], np.float32)
        sq[i].material = n_spheres + i
    lights = (api_mod.Light * 1)()
    if point_light:
        lights[0].type = 0
        lights[0].pos[:] = [-5.0, 5.0, 5.0]
    else:
        lights[0].type = 1
        lights[0].pos[:] = [0.0, 1.5, 0.0]
        lights[0].p0[:] = [-2.0, -0.5, 0.0]
        lights[0].p1[:] = [2.0, -0.5, 0.0]
        lights[0].p2[:] = [2.0, 3.5, 0.0]
    lights[0].color[:] = [1.0, 1.0, 1.0]
    d = api_mod.SceneDesc()
    d.n_spheres, d.n_squares, d.n_lights, d.n_materials = n_spheres, n_squares, 1, len(mats)
    d.spheres, d.squares, d.lights, d.materials = sph, sq, lights, mats
    keep = [mats, sph, sq, lights]
    if with_mesh is not None:
        parts = [with_mesh] + ([second_mesh] if second_mesh is not None else [])
        pos = np.ascontiguousarray(np.concatenate([p[0] for p in parts]), np.float32)
        nrm = np.ascontiguousarray(np.concatenate([p[1] for p in parts]), np.float32)
        tri = np.ascontiguousarray(np.concatenate([p[2] for p in parts]), np.uint32)   # indices stay mesh-local
        mesh = (api_mod.Mesh * len(parts))()
        v0 = t0 = 0
        for k, (p, _, t) in enumerate(parts):
            mesh[k].first_vertex, mesh[k].n_vertices = v0, len(p)
            mesh[k].first_triangle, mesh[k].n_triangles = t0, len(t)
            mesh[k].material = len(mats) - 1
            mesh[k].aabb_min[:] = np.asarray(p, np.float32).min(0)
            mesh[k].aabb_max[:] = np.asarray(p, np.float32).max(0)
            v0 += len(p)
            t0 += len(t)
        d.n_meshes, d.meshes = len(parts), mesh
        d.n_vertices, d.n_triangles = len(pos), len(tri)
        d.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
        d.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
        d.triangles = tri.ctypes.data_as(C.POINTER(C.c_uint32))
        keep += [mesh, pos, nrm, tri]
    d.kd_max_depth, d.kd_max_leaf, d.kd_epsilon = kd_max_depth, kd_max_leaf, 0.01
    return d, keep


def grid_mesh(n, seed=1, size=1.2, center=(0.0, -1.0, 0.0), bump=0.15):
    """A bumpy n x n vertex height field (2 (n-1)^2 triangles) with area-unweighted vertex normals computed
    in float32 the way Mesh::recomputeNormals does (face normals normalised, summed, normalised)."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-size, size, n, dtype=np.float32)
    gx, gz = np.meshgrid(xs, xs, indexing="xy")
    gy = (bump * rng.random((n, n))).astype(np.float32)
    pos = np.stack([gx + np.float32(center[0]), gy + np.float32(center[1]), gz + np.float32(center[2])], -1).reshape(-1, 3)
    pos = pos.astype(np.float32)
    idx = np.arange(n * n, dtype=np.uint32).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    tri = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)]).astype(np.uint32)
    e1 = pos[tri[:, 1]] - pos[tri[:, 0]]
    e2 = pos[tri[:, 2]] - pos[tri[:, 0]]
    fn = np.cross(e1, e2).astype(np.float32)
    fn /= np.sqrt((fn * fn).sum(1, dtype=np.float32), dtype=np.float32)[:, None]
    nrm = np.zeros_like(pos)
    for k in range(3):
        np.add.at(nrm, tri[:, k], fn)
    nrm /= np.sqrt((nrm * nrm).sum(1, dtype=np.float32), dtype=np.float32)[:, None]
    return pos, nrm.astype(np.float32), tri
