"""CPU tests of the host side: the product's C++ Scene / Camera classes against the oracle's own
restatement of the reference's set-up code (two independent implementations, bit-identical data),
the C-ABI library's exports, and the tile bookkeeping."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import orc
from raytracing_amd import api, host
from tests import descutil


def test_library_exports_every_declared_symbol():
    L = api.lib()
    header = open(os.path.join(os.path.dirname(api.HERE), "include", "rtamd.h")).read()
    import re
    declared = set(re.findall(r"^\s*(?:int|size_t|void|const char\*)\s+(rt_\w+)\(", header, flags=re.M))
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None


def test_device_calls_fail_loudly_without_init():
    """No CPU fallback: without rt_init (or without a GPU) every compute entry point returns an error."""
    L = api.lib()
    z = np.zeros(3, np.float32)
    out = np.zeros(1, api.HIT_DTYPE)
    rc = L.rt_intersect(1, z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p))
    assert rc in (-2, -4) and L.rt_last_error()
    rc = L.rt_scene_refit(1, z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), 1, None)
    assert rc in (-2, -4) and b"rt_scene_refit" in L.rt_last_error()


@pytest.mark.parametrize("sid", [0, 1, 2, 3])
def test_host_scene_equals_oracle_scene(sid):
    so, sh = orc.Scene.builtin(sid), host.HostScene.builtin(sid)  # keep both alive: desc points into them
    a, b = descutil.describe(so.desc), descutil.describe(sh.desc)
    for k in a:
        assert a[k] == b[k], f"scene {sid}: {k} differs"


def test_host_scene_mesh_aabb_replay():
    o = orc.Scene.builtin(3)
    o.compute_mesh_aabbs()
    a = descutil.describe(o.desc)
    sh = host.HostScene.builtin(3, compute_mesh_aabbs=True)
    b = descutil.describe(sh.desc)
    assert a["meshes"] == b["meshes"]
    assert np.isfinite(np.frombuffer(a["meshes"][0][1], np.float32)).all()


@pytest.mark.parametrize("wh", [(480, 480), (1920, 1080), (3840, 2160), (100, 37)])
def test_host_camera_equals_oracle_camera(wh):
    a, b = orc.camera_default(*wh), host.camera_default(*wh)
    assert descutil.raw(a) == descutil.raw(b)


def test_rotated_camera_equals_oracle_camera():
    q = np.array([0.1, -0.2, 0.3, 0.9], np.float32)
    q /= np.float32(np.linalg.norm(q))
    a = orc.camera_build(60.0, 1.5, 500.0, 640, 360, 0.3, -0.2, -3.1, 2.5, q)
    b = host.camera_build(60.0, 1.5, 500.0, 640, 360, 0.3, -0.2, -3.1, 2.5, q)
    assert descutil.raw(a) == descutil.raw(b)


def test_tiles_per_rank_partition():
    for (w, h, tw, th) in [(1920, 1080, 16, 16), (100, 37, 16, 8), (7, 5, 8, 8)]:
        total = ((w + tw - 1) // tw) * ((h + th - 1) // th)
        for world in (1, 2, 3, 8):
            per = [api.tiles_per_rank(w, h, tw, th, r, world) for r in range(world)]
            assert sum(per) == total and max(per) == per[0] and max(per) - min(per) <= 1
    assert api.tiles_per_rank(10, 10, 0, 8, 0, 1) == 0 and api.tiles_per_rank(10, 10, 8, 8, 2, 2) == 0


def test_tiles_to_image_roundtrip():
    w, h, tw, th, world = 50, 21, 16, 8, 3
    img = np.random.default_rng(0).random((h, w, 3)).astype(np.float32)
    tx, ty = (w + tw - 1) // tw, (h + th - 1) // th
    per = api.tiles_per_rank(w, h, tw, th, 0, world)
    buf = np.zeros((world, per, th, tw, 3), np.float32)
    for t in range(tx * ty):
        x0, y0 = ((t % tx + t // tx) % tx) * tw, (t // tx) * th  # row t // tx is rotated by its index
        blk = img[y0:y0 + th, x0:x0 + tw]
        buf[t % world, t // world, :blk.shape[0], :blk.shape[1]] = blk
    out = api.tiles_to_image(buf.reshape(world, -1), w, h, tw, th, world)
    assert np.array_equal(out, img)


def test_ppm_writer_matches_reference_format(tmp_path):
    rgb = np.array([[[0.0, 0.5, 1.0], [2.0, 0.999, 0.00392157]]], np.float32)
    p = str(tmp_path / "x.ppm")
    host.write_ppm(p, rgb)
    toks = open(p).read().split()
    assert toks[:4] == ["P3", "2", "1", "255"]
    assert [int(t) for t in toks[4:]] == list(orc.quantize(rgb).ravel())


def test_rt_write_ppm_p3_and_p6(tmp_path):
    """The C-ABI writer: P3 token for token the layout of the reference (and of host.write_ppm), P6 the same bytes."""
    rng = np.random.default_rng(3)
    rgb8 = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    p3, p6 = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    api.write_ppm(p3, rgb8, binary=False)
    api.write_ppm(p6, rgb8, binary=True)
    text = open(p3).read()
    assert text.startswith("P3\n5 7\n255\n") and text.endswith(" \n")
    assert [int(t) for t in text.split()[4:]] == list(rgb8.ravel())
    raw = open(p6, "rb").read()
    assert raw.startswith(b"P6\n5 7\n255\n") and raw[len(b"P6\n5 7\n255\n"):] == rgb8.tobytes()
    # same P3 tokens as the host writer that follows R/main.cpp:204-216
    ref = str(tmp_path / "c.ppm")
    host.write_ppm(ref, rgb8.astype(np.float32) / np.float32(255.0) + np.float32(1e-4))
    assert open(ref).read().split() == text.split()
    with pytest.raises(api.RtError):
        api.write_ppm(str(tmp_path / "no" / "such" / "dir.ppm"), rgb8)


def test_coff_file_loads_like_the_same_mesh_without_colours(tmp_path):
    """loadOFF skips the per-vertex colours of a COFF file (the reference's reader cannot load one,
    R/src/Mesh.cpp:30-31): positions, normals and triangles equal those of the plain OFF file."""
    rng = np.random.default_rng(5)
    pos = rng.uniform(-1, 1, (40, 3)).astype(np.float32)
    tri = np.array([[i % 40, (i + 1) % 40, (i + 7 + i // 40) % 40] for i in range(60)], np.uint32)
    plain, coloured = tmp_path / "m.off", tmp_path / "m_c.off"
    with open(plain, "w") as f:
        f.write(f"OFF\n{len(pos)} {len(tri)} 0\n")
        f.writelines(f"{p[0]!r} {p[1]!r} {p[2]!r}\n" for p in pos.tolist())
        f.writelines(f"3 {t[0]} {t[1]} {t[2]}\n" for t in tri)
    with open(coloured, "w") as f:
        f.write(f"COFF\n{len(pos)} {len(tri)} 0\n")
        f.writelines(f"{p[0]!r} {p[1]!r} {p[2]!r} {int(c[0])} {int(c[1])} {int(c[2])} 255\n" for p, c in zip(pos.tolist(), rng.integers(0, 256, (40, 3))))
        f.writelines(f"3 {t[0]} {t[1]} {t[2]} 0.5 0.25 1.0\n" for t in tri)
    a, b = host.HostScene.mesh_file(str(plain)), host.HostScene.mesh_file(str(coloured))
    da, db = descutil.describe(a.desc), descutil.describe(b.desc)
    assert a.desc.n_triangles == 60 and a.desc.n_vertices == 40
    for k in da:
        assert da[k] == db[k], f"{k} differs"


def _write_off(path, tokens_v, faces, tag="OFF", nv=None, nt=None):
    with open(path, "w") as f:
        f.write(f"{tag}\n{nv if nv is not None else len(tokens_v) // 3} {nt if nt is not None else len(faces)} 0\n")
        for i in range(0, len(tokens_v), 3):
            f.write(" ".join(tokens_v[i:i + 3]) + "\n")
        for fc in faces:
            f.write(fc + "\n")


def test_fast_off_reader_gives_the_token_readers_values(tmp_path):
    """Decimal formats of every shape: short ones (exact fast path), long mantissas, big and tiny exponents,
    values that round at the last bit -- loadOFF must equal the token-by-token reader bit for bit."""
    rng = np.random.default_rng(11)
    toks = []
    for _ in range(3000):
        x = float(np.float32(rng.normal() * 10.0 ** rng.integers(-12, 12)))
        fmt = rng.choice(["%.7g", "%.9g", "%.17g", "%.3f", "%e", "%.12f", "%.4e", "%d"])
        toks.append(fmt % (int(x) % 100000 if fmt == "%d" else x))
    toks += ["0", "-0", "0.0", ".5", "-.25", "5.", "1e5", "1E-5", "16777215", "16777217", "0.1", "0.30000001192092896",
             "1.17549435e-38", "3.40282347e38", "9999999", "1234567e-3", "0.000001234567", "8388608.5", "1e10", "1e-10", "1e11"]
    toks += ["0"] * (-len(toks) % 3)
    nv = len(toks) // 3
    faces = [f"3 {i % nv} {(i * 7) % nv} {(i * 13 + 1) % nv}" for i in range(500)]
    p = tmp_path / "a.off"
    _write_off(p, toks, faces)
    same, fast, n_v, n_t = host.off_compare(str(p))
    assert fast and same and (n_v, n_t) == (nv, 500)
    # a file large enough for several reader threads (> 1 MiB of text)
    big_v = ["%.7g" % v for v in rng.normal(size=3 * 40000)]
    big_f = [f"3 {i % 40000} {(i + 1) % 40000} {(i + 2) % 40000}" for i in range(80000)]
    p2 = tmp_path / "big.off"
    _write_off(p2, big_v, big_f)
    assert os.path.getsize(p2) > (1 << 20)
    assert host.off_compare(str(p2)) == (True, True, 40000, 80000)


@pytest.mark.parametrize("case", ["plus", "inf", "hex", "short", "coff", "overflow", "junk"])
def test_fast_off_reader_declines_what_it_cannot_reproduce(tmp_path, case):
    """Tokens whose treatment by operator>> is special (leading '+', inf, hex floats, range errors, junk, a file
    that ends early, COFF): the fast reader must decline, so that loadOFF behaves exactly like the token reader."""
    v = ["0.5", "1.5", "-2", "1", "2", "3", "4", "5", "6"]
    faces = ["3 0 1 2", "3 2 1 0"]
    tag, nv, nt = "OFF", None, None
    if case == "plus": v[1] = "+1.5"
    elif case == "inf": v[2] = "inf"
    elif case == "hex": v[0] = "0x1p3"
    elif case == "short": nt = 5            # header promises more faces than the file holds
    elif case == "coff": tag = "COFF"
    elif case == "overflow": v[4] = "1e40"
    elif case == "junk": v[3] = "1.5abc"
    p = tmp_path / f"{case}.off"
    _write_off(p, v, faces, tag=tag, nv=nv, nt=nt)
    same, fast, _, _ = host.off_compare(str(p))
    assert same and not fast


def test_off_file_with_an_out_of_range_face_index_is_refused(tmp_path):
    """A face that names a vertex the file does not have is refused when the file is read (message + failure exit, the
    reference's convention for an unusable mesh file) -- before recomputeNormals could write outside the vertex array and
    before anything touches a GPU.  Both readers: the multi-threaded one and the token reader (COFF input)."""
    import subprocess
    rt_main = os.path.join(os.path.dirname(os.path.abspath(host.__file__)), "host", "rt_main")
    for tag, extra in (("OFF", ""), ("COFF", " 255 0 0 255")):
        bad = tmp_path / f"bad_{tag}.off"
        bad.write_text(f"{tag}\n4 2 0\n" + "".join(f"{x} {y} 0{extra}\n" for x, y in ((0, 0), (1, 0), (1, 1), (0, 1))) + "3 0 1 2\n3 0 2 7\n")
        r = subprocess.run([rt_main, "--keys", "", "--off", str(bad)], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "refers to vertex 7 of 4" in r.stderr, (tag, r.returncode, r.stderr)
    neg = tmp_path / "neg.off"   # operator>> reads "-1" into an unsigned as 4294967295
    neg.write_text("OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 2 -1 0\n")
    r = subprocess.run([rt_main, "--keys", "", "--off", str(neg)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "refers to vertex 4294967295 of 3" in r.stderr, (r.returncode, r.stderr)


def test_triangle_count_limit():
    """2^28 triangles or more are refused with RT_E_LIMIT before anything is read (BVH leaf entries hold 28-bit positions)."""
    d = api.SceneDesc()
    d.n_triangles = 1 << 28
    d.kd_max_depth, d.kd_max_leaf, d.kd_epsilon = 15, 30, 0.01
    with pytest.raises(api.RtError) as e:
        api.scene_check(d)
    assert e.value.code == -5
