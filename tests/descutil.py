"""Semantic comparison of two rt_scene_desc structures (ctypes, any binding)."""
import ctypes as C

import numpy as np


def _arr(ptr, n, width):
    if n == 0:
        return np.zeros((0, width), np.uint32)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n * width,)).reshape(n, width).copy()


def raw(x):
    return C.string_at(C.addressof(x), C.sizeof(x))


def _mat(desc, idx):
    return raw(desc.materials[idx])


def describe(desc):
    """Canonical form: per object its raw bytes with the material index replaced by the material's bytes."""
    out = {"counts": (desc.n_spheres, desc.n_squares, desc.n_meshes, desc.n_lights, desc.n_vertices, desc.n_triangles),
           "kd": (desc.kd_max_depth, desc.kd_max_leaf, np.float32(desc.kd_epsilon).tobytes())}
    out["spheres"] = [(raw(desc.spheres[i])[:16], _mat(desc, desc.spheres[i].material)) for i in range(desc.n_spheres)]
    out["squares"] = [(raw(desc.squares[i])[:48], _mat(desc, desc.squares[i].material)) for i in range(desc.n_squares)]
    out["meshes"] = [((m.first_vertex, m.n_vertices, m.first_triangle, m.n_triangles), bytes(m.aabb_min), bytes(m.aabb_max),
                      _mat(desc, m.material)) for m in (desc.meshes[i] for i in range(desc.n_meshes))]
    out["lights"] = [raw(desc.lights[i]) for i in range(desc.n_lights)]
    out["positions"] = _arr(desc.positions, desc.n_vertices, 3).tobytes()
    out["normals"] = _arr(desc.normals, desc.n_vertices, 3).tobytes()
    out["triangles"] = _arr(desc.triangles, desc.n_triangles, 3).tobytes()
    return out
