"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librt_oracle.so")
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
STAR_OFF = os.path.join(GOLDEN, "star.off")

RT_RNG_COUNTER, RT_RNG_LIBC = 0, 1


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("shininess", C.c_float),
                ("index_medium", C.c_float), ("type", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", C.c_int32)]


class Square(C.Structure):
    _fields_ = [("v", (C.c_float * 3) * 4), ("material", C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("n_vertices", C.c_uint32), ("first_triangle", C.c_uint32),
                ("n_triangles", C.c_uint32), ("material", C.c_int32), ("aabb_min", C.c_float * 3),
                ("aabb_max", C.c_float * 3)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("pos", C.c_float * 3), ("color", C.c_float * 3), ("p0", C.c_float * 3),
                ("p1", C.c_float * 3), ("p2", C.c_float * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("n_spheres", C.c_uint32), ("n_squares", C.c_uint32), ("n_meshes", C.c_uint32),
                ("n_lights", C.c_uint32), ("n_materials", C.c_uint32),
                ("spheres", C.POINTER(Sphere)), ("squares", C.POINTER(Square)), ("meshes", C.POINTER(Mesh)),
                ("lights", C.POINTER(Light)), ("materials", C.POINTER(Material)),
                ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32),
                ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("triangles", C.POINTER(C.c_uint32)),
                ("kd_max_depth", C.c_int32), ("kd_max_leaf", C.c_int32), ("kd_epsilon", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("projection", C.c_float * 16), ("modelview", C.c_float * 16), ("depth_near", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("nodes", C.c_uint64), ("kernel_ms", C.c_double),
                ("bvh_nodes", C.c_uint64), ("bvh_tris", C.c_uint64), ("rays_skipped", C.c_uint64)]


HIT_DTYPE = np.dtype([("hit", "<i4"), ("type", "<i4"), ("object", "<i4"), ("tri", "<i4"), ("t", "<f4"),
                      ("u", "<f4"), ("v", "<f4"), ("pad0", "<f4"), ("p", "<f4", 3), ("pad1", "<f4"),
                      ("n", "<f4", 3), ("pad2", "<f4")])
assert HIT_DTYPE.itemsize == 64

COUNTER_NAMES = ["rays", "samples", "nodes", "aabb", "sph_miss", "sph_hit", "sq_behind", "sq_reject", "sq_hit",
                 "tri_par", "tri_u", "tri_v", "tri_full", "tri_upd", "leaf_refs", "light_samples"]


def build(force=False):
    """Compile the oracle with its Makefile if the library is missing or stale."""
    src = os.path.join(HERE, "rt_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "librt_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        fp, u32p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
        L.orc_scene_builtin.restype = vp
        L.orc_scene_builtin.argtypes = [C.c_int, C.c_char_p]
        L.orc_scene_from_desc.restype = vp
        L.orc_scene_from_desc.argtypes = [C.POINTER(SceneDesc)]
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_desc.restype = C.POINTER(SceneDesc)
        L.orc_scene_desc.argtypes = [vp]
        L.orc_scene_compute_mesh_aabbs.argtypes = [vp]
        L.orc_scene_kdtree.argtypes = [vp, vp, u32p, vp, u32p, fp]
        L.orc_camera_default.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Camera)]
        L.orc_camera_build.argtypes = [C.c_float] * 3 + [C.c_uint32] * 2 + [C.c_float] * 4 + [fp, C.POINTER(Camera)]
        L.orc_generate_rays.argtypes = [C.POINTER(Camera), vp, vp, C.c_size_t, vp, vp]
        L.orc_normalize_dirs.argtypes = [vp, C.c_size_t, vp]
        L.orc_intersect.argtypes = [vp, vp, vp, C.c_size_t, vp]
        L.orc_intersect_mt.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_int]
        L.orc_render_pixels.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t,
                                        C.c_int, vp, C.POINTER(Stats)]
        L.orc_primitive_intersect.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_size_t, vp]
        L.orc_mesh_intersect.argtypes = [vp, C.c_uint32, vp, vp, C.c_size_t, vp]
        L.orc_trace.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.c_uint32, C.c_uint32, vp, C.POINTER(Stats)]
        L.orc_render.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_int, vp, C.POINTER(Stats)]
        L.orc_quantize.argtypes = [vp, C.c_size_t, vp]
        L.orc_scene_counters.restype = C.c_double
        L.orc_scene_counters.argtypes = [vp, vp]
        L.orc_scene_counters_reset.argtypes = [vp]
        L.orc_scene_discarded_work.restype = C.c_double
        L.orc_scene_discarded_work.argtypes = [vp, vp]
        L.orc_selfcheck.restype = C.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Scene:
    """An oracle scene: one of the reference's four built-in scenes, or a copy of a descriptor."""

    def __init__(self, handle, keep=None):
        if not handle:
            raise RuntimeError("oracle scene construction failed")
        self.h = handle
        self._keep = keep

    @classmethod
    def builtin(cls, scene_id, off_path=STAR_OFF):
        return cls(lib().orc_scene_builtin(scene_id, off_path.encode()))

    @classmethod
    def from_desc(cls, desc, keep=None):
        return cls(lib().orc_scene_from_desc(C.byref(desc)), keep)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_free(self.h)
            self.h = None

    @property
    def desc(self):
        return lib().orc_scene_desc(self.h).contents

    def compute_mesh_aabbs(self):
        lib().orc_scene_compute_mesh_aabbs(self.h)

    def kdtree(self):
        nn, nr = C.c_uint32(), C.c_uint32()
        box = np.zeros(6, np.float32)
        lib().orc_scene_kdtree(self.h, None, C.byref(nn), None, C.byref(nr), box.ctypes.data_as(C.POINTER(C.c_float)))
        nodes = np.zeros((nn.value, 8), np.int32)
        refs = np.zeros((nr.value, 2), np.int32)
        lib().orc_scene_kdtree(self.h, _ptr(nodes), C.byref(nn), _ptr(refs), C.byref(nr),
                               box.ctypes.data_as(C.POINTER(C.c_float)))
        return nodes, refs, box

    def intersect(self, org, dirs, threads=1):
        org, dirs = _f32(org), _f32(dirs)
        n = org.shape[0]
        out = np.zeros(n, HIT_DTYPE)
        if threads > 1:
            lib().orc_intersect_mt(self.h, _ptr(org), _ptr(dirs), n, _ptr(out), threads)
        else:
            lib().orc_intersect(self.h, _ptr(org), _ptr(dirs), n, _ptr(out))
        return out

    def render_pixels(self, cam, w, h, spp, pixels, seed=1, threads=1):
        """Colours of the listed pixels (x + y * w) of a w x h render under the counter RNG: (n, 3) float32, Stats."""
        pixels = np.ascontiguousarray(pixels, np.uint32)
        rgb = np.zeros((len(pixels), 3), np.float32)
        st = Stats()
        lib().orc_render_pixels(self.h, C.byref(cam), w, h, spp, seed, _ptr(pixels), len(pixels), threads, _ptr(rgb), C.byref(st))
        return rgb, st

    def primitive_intersect(self, type_, obj, org, dirs):
        """Sphere::intersect (type 1) / Square::intersect (type 2) of object `obj` alone."""
        org, dirs = _f32(org), _f32(dirs)
        n = org.shape[0]
        out = np.zeros(n, HIT_DTYPE)
        lib().orc_primitive_intersect(self.h, type_, obj, _ptr(org), _ptr(dirs), n, _ptr(out))
        return out

    def mesh_intersect(self, mesh, org, dirs):
        org, dirs = _f32(org), _f32(dirs)
        n = org.shape[0]
        out = np.zeros(n, HIT_DTYPE)
        lib().orc_mesh_intersect(self.h, mesh, _ptr(org), _ptr(dirs), n, _ptr(out))
        return out

    def trace(self, org, dirs, pixel, sample, seed=1, k0=2):
        org, dirs = _f32(org), _f32(dirs)
        pixel = np.ascontiguousarray(pixel, np.uint32)
        sample = np.ascontiguousarray(sample, np.uint32)
        n = org.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        st = Stats()
        lib().orc_trace(self.h, _ptr(org), _ptr(dirs), n, _ptr(pixel), _ptr(sample), seed, k0, _ptr(rgb), C.byref(st))
        return rgb, st

    def render(self, cam, w, h, spp, rng_mode=RT_RNG_COUNTER, seed=1, y0=0, y1=None, threads=1):
        y1 = h if y1 is None else y1
        rgb = np.zeros((h, w, 3), np.float32)
        st = Stats()
        lib().orc_render(self.h, C.byref(cam), w, h, spp, rng_mode, seed, y0, y1, threads, _ptr(rgb), C.byref(st))
        return rgb, st

    def counters(self):
        out = np.zeros(16, np.uint64)
        flops = lib().orc_scene_counters(self.h, _ptr(out))
        return dict(zip(COUNTER_NAMES, (int(v) for v in out))), flops

    def discarded_work(self):
        """(closest-hit queries, algorithmic FLOP) the reference spends on Phong colours it overwrites (glass nodes)."""
        rays = np.zeros(1, np.uint64)
        flops = lib().orc_scene_discarded_work(self.h, _ptr(rays))
        return int(rays[0]), flops

    def reset_counters(self):
        lib().orc_scene_counters_reset(self.h)


def camera_default(w, h):
    cam = Camera()
    lib().orc_camera_default(w, h, C.byref(cam))
    return cam


def camera_build(fov, z_near, z_far, w, h, x, y, z, zoom, quat):
    cam = Camera()
    q = (C.c_float * 4)(*quat)
    lib().orc_camera_build(fov, z_near, z_far, w, h, x, y, z, zoom, q, C.byref(cam))
    return cam


def generate_rays(cam, u, v):
    u, v = _f32(u), _f32(v)
    n = u.shape[0]
    org = np.zeros((n, 3), np.float32)
    dirs = np.zeros((n, 3), np.float32)
    lib().orc_generate_rays(C.byref(cam), _ptr(u), _ptr(v), n, _ptr(org), _ptr(dirs))
    return org, dirs


def normalize_dirs(dirs):
    """Ray-constructor normalisation of raw directions (float32, the reference's operation order)."""
    dirs = _f32(dirs)
    out = np.zeros_like(dirs)
    lib().orc_normalize_dirs(_ptr(dirs), dirs.shape[0], _ptr(out))
    return out


def quantize(rgb):
    rgb = _f32(rgb)
    out = np.zeros(rgb.shape, np.uint8)
    lib().orc_quantize(_ptr(rgb), rgb.size, _ptr(out))
    return out
