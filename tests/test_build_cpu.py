"""Host-side scene preparation without a GPU (rt_scene_check): the triangle BVH the device traverses is
built and verified on the CPU -- every triangle in exactly one leaf, boxes nested and enclosing their
triangles, no path longer than the device's 32-entry traversal stack -- for ordinary, degenerate and
adversarial meshes, and the description checks of rt_scene_create answer the same way."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import api
from tests import scenegen


def mesh_scene(pos, tri):
    nrm = np.zeros_like(pos)
    nrm[:, 1] = 1.0
    return scenegen.random_objects(api, n_spheres=2, n_squares=2, seed=1, with_mesh=(pos, nrm, tri))


def test_grid_mesh_bvh_is_verified():
    pos, nrm, tri = scenegen.grid_mesh(160, seed=2)
    d, keep = scenegen.random_objects(api, n_spheres=3, n_squares=3, seed=5, with_mesh=(pos, nrm, tri))
    info = api.scene_check(d)
    assert info.bvh_meshes == 1 and info.kernel_variant & 2
    assert info.bvh_nodes >= len(tri) // 4 and 10 <= info.bvh_max_depth <= 32
    assert info.device_bytes > len(tri) * 48


def test_small_mesh_keeps_the_brute_force_loop():
    pos, nrm, tri = scenegen.grid_mesh(5, seed=2)  # 32 triangles <= 64
    d, keep = scenegen.random_objects(api, n_spheres=3, n_squares=3, seed=5, with_mesh=(pos, nrm, tri))
    info = api.scene_check(d)
    assert info.bvh_meshes == 0 and info.bvh_nodes == 0 and info.kernel_variant == 0


def test_many_objects_give_kd_internal_nodes():
    d, keep = scenegen.random_objects(api, n_spheres=60, n_squares=30, seed=7)
    info = api.scene_check(d)
    assert info.kernel_variant == 1 and info.kd_nodes > 1 and 1 <= info.kd_max_depth < 15


def test_identical_triangles_fall_back_to_median_splits():
    """Centroids coincide: the SAH has nothing to split on, the median split still halves the range."""
    n = 5000
    pos = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1]], np.float32), (n, 1))
    tri = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    d, keep = mesh_scene(pos, tri)
    info = api.scene_check(d)
    assert info.bvh_meshes == 1 and info.bvh_max_depth <= 12  # log2(5000 / 4) rounded up + 1


def test_geometric_progression_stays_within_the_stack():
    """Triangle k is 1.17 x larger and further out than triangle k-1 (sizes over 27 orders of magnitude): the
    kind of input that makes a surface-area split peel off a few triangles per level."""
    n = 400
    s = (1.17 ** np.arange(n)).astype(np.float64)
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1]], np.float64)
    pos = (base[None, :, :] * s[:, None, None] + np.stack([s * 3, 0 * s, 0 * s], -1)[:, None, :]).reshape(-1, 3).astype(np.float32)
    tri = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    d, keep = mesh_scene(pos, tri)
    info = api.scene_check(d)
    assert info.bvh_meshes == 1 and info.bvh_max_depth <= 32


def test_random_soup_200k():
    rng = np.random.default_rng(4)
    n = 200_000
    c = rng.uniform(-1, 1, (n, 1, 3))
    pos = (c + rng.normal(scale=0.01, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    tri = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    d, keep = mesh_scene(pos, tri)
    info = api.scene_check(d)
    assert info.bvh_meshes == 1 and info.bvh_max_depth <= 32 and info.bvh_nodes < n


def test_description_errors_match_scene_create():
    pos, nrm, tri = scenegen.grid_mesh(12, seed=2)
    bad = tri.copy()
    bad[7, 1] = len(pos) + 3  # vertex index outside the mesh
    d, keep = scenegen.random_objects(api, n_spheres=2, n_squares=2, seed=1, with_mesh=(pos, nrm, bad))
    with pytest.raises(api.RtError, match="vertex index out of range"):
        api.scene_check(d)
    d, keep = scenegen.random_objects(api, n_spheres=2, n_squares=2, seed=1)
    d.spheres[0].material = 99
    with pytest.raises(api.RtError, match="material index out of range"):
        api.scene_check(d)
    with pytest.raises(api.RtError):
        api.lib().rt_scene_check.restype = C.c_int
        api.check(api.lib().rt_scene_check(None, None))


def test_builder_under_address_and_ub_sanitizers(tmp_path):
    """rt_build.cpp (validation, kd-tree, BVH, camera inversion) compiled for the host with ASan + UBSan and run
    on ordinary, degenerate (NaN / huge coordinates, identical triangles) and invalid descriptions."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "build_harness")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-pthread",
           "-I", os.path.join(root, "raytracing_amd", "csrc"), "-I", os.path.join(root, "include"),
           os.path.join(root, "tests", "sanitize", "build_harness.cpp"), os.path.join(root, "raytracing_amd", "csrc", "rt_build.cpp"),
           "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if built.returncode != 0 and "sanitize" in built.stderr and "cannot find" in built.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert built.returncode == 0, built.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and run.stdout.strip().startswith("OK"), (run.stdout + run.stderr)[-3000:]
