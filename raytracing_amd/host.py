"""Python view of the C++ host classes (raytracing_amd/host/*.hpp through librtamd_host.so):
scene set-up and camera state exactly as the product's C++ host code computes them."""
import ctypes as C
import os

import numpy as np

from . import api

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "host", "librtamd_host.so")
ASSETS = os.path.join(HERE, "assets")
STAR_OFF = os.path.join(ASSETS, "img", "star.off")

_lib = None


def lib():
    global _lib
    if _lib is None:
        api.lib()  # dependency, and the loud failure if the HIP library is missing
        if not os.path.exists(LIB_PATH):
            raise api.RtError(-3, f"{LIB_PATH} not found: build it with `make -C raytracing_amd/host`")
        L = C.CDLL(LIB_PATH)
        L.rth_scene_builtin.restype = C.c_void_p
        L.rth_scene_builtin.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.rth_scene_mesh_file.restype = C.c_void_p
        L.rth_scene_mesh_file.argtypes = [C.c_char_p, C.c_int]
        L.rth_scene_desc.restype = C.POINTER(api.SceneDesc)
        L.rth_scene_desc.argtypes = [C.c_void_p]
        L.rth_scene_free.argtypes = [C.c_void_p]
        L.rth_camera_default.argtypes = [C.c_uint, C.c_uint, C.POINTER(api.Camera)]
        L.rth_camera_build.argtypes = [C.c_float] * 3 + [C.c_uint] * 2 + [C.c_float] * 4 + [C.POINTER(C.c_float),
                                                                                          C.POINTER(api.Camera)]
        L.rth_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint, C.c_uint]
        L.rth_off_compare.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        _lib = L
    return _lib


class HostScene:
    """A scene built by the C++ Scene class and flattened (no GPU involved)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("host scene construction failed")
        self.h = handle

    @classmethod
    def builtin(cls, scene_id, off_path=STAR_OFF, compute_mesh_aabbs=False):
        return cls(lib().rth_scene_builtin(scene_id, off_path.encode(), int(compute_mesh_aabbs)))

    @classmethod
    def mesh_file(cls, off_path, with_spheres=False):
        return cls(lib().rth_scene_mesh_file(off_path.encode(), int(with_spheres)))

    @property
    def desc(self):
        return lib().rth_scene_desc(self.h).contents

    def upload(self, flags=None):
        """rt_scene_create (flags None) or rt_scene_create_ex on the current GPU."""
        return api.DeviceScene(self.desc, keep=self, flags=flags)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # module globals are gone at interpreter exit
            lib().rth_scene_free(self.h)
            self.h = None


def off_compare(path):
    """Mesh::loadOFF against the reference's token-by-token reader on one file:
    (identical, fast_reader_accepted, n_vertices, n_triangles)."""
    nv, nt = C.c_uint(), C.c_uint()
    r = lib().rth_off_compare(os.fsencode(path), C.byref(nv), C.byref(nt))
    return bool(r & 1), bool(r & 2), nv.value, nt.value


def camera_default(w, h):
    cam = api.Camera()
    lib().rth_camera_default(w, h, C.byref(cam))
    return cam


def camera_build(fov, z_near, z_far, w, h, x, y, z, zoom, quat):
    cam = api.Camera()
    q = (C.c_float * 4)(*quat)
    lib().rth_camera_build(fov, z_near, z_far, w, h, x, y, z, zoom, q, C.byref(cam))
    return cam


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    if lib().rth_write_ppm(path.encode(), rgb.ctypes.data_as(C.c_void_p), w, h) != 0:
        raise IOError(f"could not write {path}")
