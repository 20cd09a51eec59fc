"""ctypes binding of librtamd.so (include/rtamd.h).  Fails loudly when the library is absent."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTAMD_LIB", os.path.join(HERE, "csrc", "librtamd.so"))  # RTAMD_LIB: A/B builds

RT_MESH, RT_SPHERE, RT_SQUARE = 0, 1, 2
RT_MAT_DIFFUSE, RT_MAT_GLASS, RT_MAT_MIRROR = 0, 1, 2
RT_LIGHT_SPHERICAL, RT_LIGHT_QUAD = 0, 1
COUNTER_SLOTS, COUNTER_STRIDE = 1024, 8  # RT_COUNTER_SLOTS / RT_COUNTER_STRIDE of include/rtamd.h


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtamd error {code}: {msg}")
        self.code = code


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


class ShardedTimes(C.Structure):
    _fields_ = [("render_ms", C.c_double), ("gather_ms", C.c_double), ("untile_ms", C.c_double), ("copy_ms", C.c_double),
                ("host_copy_ms", C.c_double), ("total_ms", C.c_double)]


class BuildInfo(C.Structure):
    _fields_ = [("kd_nodes", C.c_uint32), ("kd_max_depth", C.c_uint32), ("bvh_meshes", C.c_uint32), ("bvh_nodes", C.c_uint32),
                ("bvh_max_depth", C.c_uint32), ("kernel_variant", C.c_uint32), ("device_bytes", C.c_uint64)]


class BuildTimes(C.Structure):
    _fields_ = [("host_ms", C.c_double), ("device_sort_ms", C.c_double), ("device_tree_ms", C.c_double), ("total_ms", C.c_double)]


BUILD_BVH_DEVICE = 1

HIT_DTYPE = np.dtype([("hit", "<i4"), ("type", "<i4"), ("object", "<i4"), ("tri", "<i4"), ("t", "<f4"),
                      ("u", "<f4"), ("v", "<f4"), ("pad0", "<f4"), ("p", "<f4", 3), ("pad1", "<f4"),
                      ("n", "<f4", 3), ("pad2", "<f4")])

# every symbol include/rtamd.h declares
EXPORTS = ["rt_init", "rt_set_option", "rt_shutdown", "rt_last_error", "rt_scene_check", "rt_scene_create", "rt_scene_create_ex", "rt_scene_verify", "rt_scene_refit", "rt_scene_destroy", "rt_scene_kdtree",
           "rt_intersect", "rt_intersect_device", "rt_mesh_intersect", "rt_primitive_intersect", "rt_comm_unique_id", "rt_comm_init",
           "rt_comm_destroy", "rt_comm_rank", "rt_comm_world", "rt_gather_tiles_device", "rt_render_sharded", "rt_render_sharded_device", "rt_sharded_last_times", "rt_device_count", "rt_generate_rays", "rt_trace",
           "rt_tiles_per_rank", "rt_render", "rt_render_device", "rt_render_pass_device", "rt_render_device_noalloc", "rt_tiles_to_image", "rt_tiles_to_rgb8_device", "rt_write_ppm", "rt_counters_total", "rt_selftest_arith", "rt_normalize_dirs"]

_lib = None


def sharded_last_times():
    t = ShardedTimes()
    check(lib().rt_sharded_last_times(C.byref(t)))
    return t


def device_count():
    n = C.c_int(0)
    check(lib().rt_device_count(C.byref(n)))
    return n.value


def lib():
    """Load librtamd.so; raise if it was not built (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(-3, f"{LIB_PATH} not found: build it with `make -C raytracing_amd/csrc` "
                              "(or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.rt_last_error.restype = C.c_char_p
        L.rt_init.argtypes = [C.c_int]
        L.rt_set_option.argtypes = [C.c_char_p, C.c_int64]
        L.rt_scene_check.argtypes = [C.POINTER(SceneDesc), C.POINTER(BuildInfo)]
        L.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(u64)]
        L.rt_scene_create_ex.argtypes = [C.POINTER(SceneDesc), u32, C.POINTER(u64)]
        L.rt_scene_verify.argtypes = [u64, C.POINTER(BuildInfo), C.POINTER(BuildTimes)]
        L.rt_scene_refit.argtypes = [u64, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
        L.rt_scene_destroy.argtypes = [u64]
        L.rt_scene_kdtree.argtypes = [u64, vp, C.POINTER(u32), vp, C.POINTER(u32), vp]
        L.rt_intersect.argtypes = [u64, vp, vp, C.c_size_t, vp]
        L.rt_intersect_device.argtypes = [u64, vp, vp, C.c_size_t, vp, vp]
        L.rt_mesh_intersect.argtypes = [u64, u32, vp, vp, C.c_size_t, vp]
        L.rt_primitive_intersect.argtypes = [u64, C.c_int, u32, vp, vp, C.c_size_t, vp]
        L.rt_comm_unique_id.argtypes = [vp]
        L.rt_comm_init.argtypes = [vp, C.c_int, C.c_int]
        L.rt_gather_tiles_device.argtypes = [vp, C.c_size_t, vp, C.c_int, vp]
        L.rt_render_sharded.argtypes = [u64, C.POINTER(Camera)] + [u32] * 6 + [vp, C.POINTER(Stats)]
        L.rt_render_sharded_device.argtypes = [u64, C.POINTER(Camera)] + [u32] * 6 + [vp, vp, vp]
        L.rt_sharded_last_times.argtypes = [C.POINTER(ShardedTimes)]
        L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.rt_generate_rays.argtypes = [C.POINTER(Camera), vp, vp, C.c_size_t, vp, vp]
        L.rt_trace.argtypes = [u64, vp, vp, C.c_size_t, vp, vp, u32, u32, vp, C.POINTER(Stats)]
        L.rt_tiles_per_rank.restype = C.c_size_t
        L.rt_tiles_per_rank.argtypes = [u32] * 6
        L.rt_render.argtypes = [u64, C.POINTER(Camera)] + [u32] * 8 + [vp, C.POINTER(Stats)]
        L.rt_render_device.argtypes = [u64, C.POINTER(Camera)] + [u32] * 8 + [vp, vp, vp]
        L.rt_render_device_noalloc.argtypes = [u64, C.POINTER(Camera)] + [u32] * 8 + [vp, vp, vp]
        L.rt_render_pass_device.argtypes = [u64, C.POINTER(Camera)] + [u32] * 10 + [vp, vp, vp]
        L.rt_tiles_to_image.argtypes = [vp, u32, u32, u32, u32, u32, vp]
        L.rt_tiles_to_rgb8_device.argtypes = [vp, u32, u32, u32, u32, u32, vp, vp]
        L.rt_write_ppm.argtypes = [C.c_char_p, vp, u32, u32, C.c_int]
        L.rt_selftest_arith.argtypes = [u64, u32, C.POINTER(u64)]
        L.rt_normalize_dirs.argtypes = [vp, C.c_size_t, vp]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))


def init(device=0):
    check(lib().rt_init(device))


def set_option(name, value):
    """rt_set_option: scheduling switches ("tile_order", "wide", "wide_nodes", "treelet_order"); results never depend on them."""
    check(lib().rt_set_option(name.encode(), int(value)))


def comm_unique_id():
    """128 bytes naming a new communicator (made by one process, handed to the others)."""
    buf = C.create_string_buffer(128)
    check(lib().rt_comm_unique_id(buf))
    return buf.raw


def comm_init(id_bytes, rank, world):
    check(lib().rt_comm_init(C.create_string_buffer(bytes(id_bytes), 128), rank, world))


def comm_destroy():
    check(lib().rt_comm_destroy())


def gather_tiles_device(d_mine_ptr, floats_per_rank, d_gathered_ptr, root, stream_ptr):
    check(lib().rt_gather_tiles_device(C.c_void_p(d_mine_ptr), floats_per_rank, C.c_void_p(d_gathered_ptr), root, C.c_void_p(stream_ptr)))


def shutdown():
    check(lib().rt_shutdown())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def copy_struct(dst_type, src):
    """Convert a ctypes struct of identical layout (e.g. one produced by another binding)."""
    assert C.sizeof(dst_type) == C.sizeof(src)
    out = dst_type()
    C.memmove(C.byref(out), C.byref(src), C.sizeof(src))
    return out


class DeviceScene:
    """A scene uploaded with rt_scene_create."""

    def __init__(self, desc, keep=None, flags=None):
        """flags: None = rt_scene_create (host BVH build unless RTAMD_BVH_BUILD=device), else rt_scene_create_ex."""
        h = C.c_uint64()
        d = desc if isinstance(desc, SceneDesc) else copy_struct(SceneDesc, desc)
        if flags is None:
            check(lib().rt_scene_create(C.byref(d), C.byref(h)))
        else:
            check(lib().rt_scene_create_ex(C.byref(d), flags, C.byref(h)))
        self.h = h.value
        self._keep = keep

    def verify(self):
        """rt_scene_verify: reads the device tables back and checks the BVHs; returns (BuildInfo, BuildTimes)."""
        info, times = BuildInfo(), BuildTimes()
        check(lib().rt_scene_verify(self.h, C.byref(info), C.byref(times)))
        return info, times

    def refit(self, positions, normals):
        """rt_scene_refit: new vertex positions / normals (n_vertices x 3 float32 each), same triangles; returns the GPU ms."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        assert pos.shape == nrm.shape
        ms = C.c_float(0.0)
        check(lib().rt_scene_refit(self.h, pos.ctypes.data, nrm.ctypes.data, len(pos), C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            lib().rt_scene_destroy(self.h)
            self.h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kdtree(self):
        nn, nr = C.c_uint32(), C.c_uint32()
        box = np.zeros(6, np.float32)
        check(lib().rt_scene_kdtree(self.h, None, C.byref(nn), None, C.byref(nr), _ptr(box)))
        nodes = np.zeros((nn.value, 8), np.int32)
        refs = np.zeros((nr.value, 2), np.int32)
        check(lib().rt_scene_kdtree(self.h, _ptr(nodes), C.byref(nn), _ptr(refs), C.byref(nr), _ptr(box)))
        return nodes, refs, box

    def intersect(self, org, dirs):
        org, dirs = _f32(org), _f32(dirs)
        n = org.shape[0]
        out = np.zeros(n, HIT_DTYPE)
        check(lib().rt_intersect(self.h, _ptr(org), _ptr(dirs), n, _ptr(out)))
        return out

    def mesh_intersect(self, mesh, org, dirs):
        org, dirs = _f32(org), _f32(dirs)
        n = org.shape[0]
        out = np.zeros(n, HIT_DTYPE)
        check(lib().rt_mesh_intersect(self.h, mesh, _ptr(org), _ptr(dirs), n, _ptr(out)))
        return out

    def primitive_intersect(self, type_, obj, org, dirs):
        """rt_primitive_intersect: Sphere::intersect / Square::intersect / Mesh::intersect of one object alone."""
        org, dirs = _f32(org), _f32(dirs)
        n = org.shape[0]
        out = np.zeros(n, HIT_DTYPE)
        check(lib().rt_primitive_intersect(self.h, type_, obj, _ptr(org), _ptr(dirs), n, _ptr(out)))
        return out

    def render_sharded(self, cam, w, h, spp, seed=1, tile_w=16, tile_h=16):
        """rt_render_sharded: collective over the processes of the communicator (rank 0 of 1 without one); (h, w, 3) on rank 0."""
        cam = cam if isinstance(cam, Camera) else copy_struct(Camera, cam)
        img = np.zeros((h, w, 3), np.float32)
        st = Stats()
        check(lib().rt_render_sharded(self.h, C.byref(cam), w, h, spp, seed, tile_w, tile_h, _ptr(img), C.byref(st)))
        return img, st

    def render_sharded_device(self, cam, w, h, spp, seed, tile_w, tile_h, d_image, d_counters=0, stream=0):
        """rt_render_sharded_device: render + exchange + un-tile enqueued on `stream`; d_image: device pointer of w * h * 3 floats (rank 0)."""
        cam = cam if isinstance(cam, Camera) else copy_struct(Camera, cam)
        check(lib().rt_render_sharded_device(self.h, C.byref(cam), w, h, spp, seed, tile_w, tile_h, d_image, d_counters, stream))

    def trace(self, org, dirs, pixel, sample, seed=1, k0=2):
        org, dirs = _f32(org), _f32(dirs)
        pixel = np.ascontiguousarray(pixel, np.uint32)
        sample = np.ascontiguousarray(sample, np.uint32)
        n = org.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        st = Stats()
        check(lib().rt_trace(self.h, _ptr(org), _ptr(dirs), n, _ptr(pixel), _ptr(sample), seed, k0, _ptr(rgb),
                             C.byref(st)))
        return rgb, st

    def render_tiles(self, cam, w, h, spp, seed=1, tile_w=16, tile_h=16, rank=0, world=1):
        """Host-buffer render of this rank's tiles (tile-major layout of rt_render)."""
        cam = cam if isinstance(cam, Camera) else copy_struct(Camera, cam)
        n_tiles = lib().rt_tiles_per_rank(w, h, tile_w, tile_h, rank, world)
        out = np.zeros((n_tiles, tile_h, tile_w, 3), np.float32)
        st = Stats()
        check(lib().rt_render(self.h, C.byref(cam), w, h, spp, seed, tile_w, tile_h, rank, world, _ptr(out),
                              C.byref(st)))
        return out, st

    def render(self, cam, w, h, spp, seed=1, tile_w=16, tile_h=16):
        """Whole image on this GPU, row-major (h, w, 3) float32."""
        tiles, st = self.render_tiles(cam, w, h, spp, seed, tile_w, tile_h, 0, 1)
        return tiles_to_image(tiles.reshape(1, -1), w, h, tile_w, tile_h, 1), st

    def render_device(self, cam, w, h, spp, seed, tile_w, tile_h, rank, world, d_out_ptr, d_counters_ptr, stream_ptr):
        """Enqueue the render on a caller-owned device buffer / HIP stream (raw pointers as ints)."""
        cam = cam if isinstance(cam, Camera) else copy_struct(Camera, cam)
        check(lib().rt_render_device(self.h, C.byref(cam), w, h, spp, seed, tile_w, tile_h, rank, world,
                                     C.c_void_p(d_out_ptr), C.c_void_p(d_counters_ptr), C.c_void_p(stream_ptr)))


    def render_device_noalloc(self, cam, w, h, spp, seed, tile_w, tile_h, rank, world, d_out_ptr, d_counters_ptr, stream_ptr):
        """rt_render_device_noalloc: same result as render_device, allocates nothing (static device array)."""
        cam = cam if isinstance(cam, Camera) else copy_struct(Camera, cam)
        check(lib().rt_render_device_noalloc(self.h, C.byref(cam), w, h, spp, seed, tile_w, tile_h, rank, world,
                                             C.c_void_p(d_out_ptr), C.c_void_p(d_counters_ptr), C.c_void_p(stream_ptr)))

    def render_pass_device(self, cam, w, h, spp_pass, sample0, spp_total, seed, tile_w, tile_h, rank, world, d_out_ptr,
                           d_counters_ptr, stream_ptr):
        """One pass of a progressive render: samples [sample0, sample0 + spp_pass) on top of the sums in the buffer."""
        cam = cam if isinstance(cam, Camera) else copy_struct(Camera, cam)
        check(lib().rt_render_pass_device(self.h, C.byref(cam), w, h, spp_pass, sample0, spp_total, seed, tile_w, tile_h, rank,
                                          world, C.c_void_p(d_out_ptr), C.c_void_p(d_counters_ptr), C.c_void_p(stream_ptr)))


def tiles_to_rgb8_device(d_gathered_ptr, w, h, tile_w, tile_h, world, d_rgb8_ptr, stream_ptr):
    """Un-tile + quantise on the device (raw pointers as ints), enqueued on the stream."""
    check(lib().rt_tiles_to_rgb8_device(C.c_void_p(d_gathered_ptr), w, h, tile_w, tile_h, world, C.c_void_p(d_rgb8_ptr),
                                        C.c_void_p(stream_ptr)))


def write_ppm(path, rgb8, binary=True):
    """rgb8: (h, w, 3) uint8 -> P6 (binary) or the reference's ASCII P3 layout."""
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    check(lib().rt_write_ppm(os.fsencode(path), _ptr(rgb8), rgb8.shape[1], rgb8.shape[0], 1 if binary else 0))


def scene_check(desc):
    """Host-only dry run of rt_scene_create (no GPU): builds and verifies kd-tree and BVHs, returns BuildInfo."""
    info = BuildInfo()
    check(lib().rt_scene_check(C.byref(desc), C.byref(info)))
    return info


def normalize_dirs(dirs):
    """What the reference's Ray constructor does to a direction (host arithmetic)."""
    dirs = _f32(dirs)
    out = np.zeros_like(dirs)
    lib().rt_normalize_dirs(_ptr(dirs), dirs.shape[0], _ptr(out))
    return out


def selftest_arith(n, seed=1):
    """Number of normalisations (out of n random vectors) where the shortened divide / sqrt sequences differ
    from the IEEE operators on the device; must be 0."""
    bad = C.c_uint64()
    check(lib().rt_selftest_arith(n, seed, C.byref(bad)))
    return bad.value


def tiles_per_rank(w, h, tile_w, tile_h, rank, world):
    return lib().rt_tiles_per_rank(w, h, tile_w, tile_h, rank, world)


def tiles_to_image(gathered, w, h, tile_w, tile_h, world):
    """gathered: (world, max_tiles_per_rank * tile_h * tile_w * 3) float32 -> (h, w, 3)."""
    gathered = _f32(gathered)
    img = np.zeros((h, w, 3), np.float32)
    check(lib().rt_tiles_to_image(_ptr(gathered), w, h, tile_w, tile_h, world, _ptr(img)))
    return img


def generate_rays(cam, u, v):
    cam = cam if isinstance(cam, Camera) else copy_struct(Camera, cam)
    u, v = _f32(u), _f32(v)
    n = u.shape[0]
    org = np.zeros((n, 3), np.float32)
    dirs = np.zeros((n, 3), np.float32)
    check(lib().rt_generate_rays(C.byref(cam), _ptr(u), _ptr(v), n, _ptr(org), _ptr(dirs)))
    return org, dirs
