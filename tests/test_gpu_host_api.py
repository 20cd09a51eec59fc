"""GPU tests of the C++ host layer as a drop-in: the reference-style single-ray calls
(Scene::computeIntersection, Scene::rayTrace, Mesh::intersect) and the headless '+' / 'r' driver, each
compared with the CPU oracle.  The C++ programs are built by raytracing_amd/host/Makefile."""
import os
import subprocess

import numpy as np
import pytest

from oracle import orc
from raytracing_amd import host
from tests import rayset

pytestmark = pytest.mark.gpu
HOST_DIR = os.path.join(os.path.dirname(os.path.abspath(host.__file__)), "host")


def _hexf(tok):
    return np.float32(float.fromhex(tok)) if tok not in ("nan", "-nan", "inf", "-inf") else np.float32(tok)


@pytest.mark.parametrize("sid", [2, 3])
def test_single_ray_calls_match_oracle(sid, tmp_path):
    org, d = rayset.primary_grid(orc, 12, 10, seed=sid)
    o2, d2 = rayset.interior_rays(60, seed=sid + 40)
    org, d = np.concatenate([org, o2]), np.concatenate([d, d2])
    n = len(org)
    pix = np.arange(n, dtype=np.uint32) * 7
    smp = np.arange(n, dtype=np.uint32) % 5
    rays = tmp_path / "rays.txt"
    with open(rays, "w") as f:
        for i in range(n):
            f.write(" ".join(repr(float(x)) for x in (*org[i], *d[i])) + f" {pix[i]} {smp[i]}\n")
    out = subprocess.run([os.path.join(HOST_DIR, "api_check"), str(sid), host.STAR_OFF, str(rays)], check=True,
                         capture_output=True, text=True, timeout=300).stdout
    lines = [ln for ln in out.splitlines() if "|" in ln]
    assert len(lines) == n
    o = orc.Scene.builtin(sid)
    du = orc.normalize_dirs(d)  # what the Ray objects built by api_check hold
    hits = o.intersect(org, du)
    cols, _ = o.trace(org, du, pix, smp, seed=1, k0=2)
    mh = o.mesh_intersect(0, org, du) if sid == 3 else None
    own = {"S": o.primitive_intersect(1, 0, org, du) if o.desc.n_spheres else None,
           "Q": o.primitive_intersect(2, 0, org, du) if o.desc.n_squares else None}
    for i, ln in enumerate(lines):
        parts = [p.split() for p in ln.split("|")]
        a = parts[0]
        assert int(a[0]) == hits["hit"][i] and int(a[1]) == hits["type"][i] and int(a[2]) == hits["object"][i]
        if hits["hit"][i]:
            got = np.array([_hexf(t) for t in a[3:10]], np.float32)
            want = np.concatenate([[hits["t"][i]], hits["p"][i], hits["n"][i]]).astype(np.float32)
            assert got.tobytes() == want.tobytes(), (i, got, want)
        c = np.array([_hexf(t) for t in parts[1]], np.float32)
        assert np.abs(c - cols[i]).max() <= 2e-6
        if mh is not None:
            m = parts[2]
            assert int(m[0]) == mh["hit"][i]
            if mh["hit"][i]:
                assert int(m[1]) == mh["tri"][i] and _hexf(m[2]).tobytes() == mh["t"][i].tobytes()
                w = np.array([_hexf(t) for t in m[3:6]], np.float32)
                assert w[1].tobytes() == mh["u"][i].tobytes() and w[2].tobytes() == mh["v"][i].tobytes()
        for part in parts[2:]:   # Sphere::intersect / Square::intersect of object 0 alone
            if part[0] in own:
                want = own[part[0]]
                assert want is not None and int(part[1]) == want["hit"][i]
                if want["hit"][i]:
                    got = np.array([_hexf(t) for t in part[2:9]], np.float32)
                    ref = np.concatenate([[want["t"][i]], want["p"][i], want["n"][i]]).astype(np.float32)
                    assert got.tobytes() == ref.tobytes(), (part[0], i)
        assert sum(1 for part in parts[2:] if part[0] in own) == sum(v is not None for v in own.values())


def test_primitive_intersect_and_sharded_render_of_one_rank():
    """rt_primitive_intersect over many rays against the oracle's Sphere::intersect / Square::intersect, bit for bit (rays
    from inside the sphere, tangent ones, rays in the square's plane); rt_render_sharded with a one-rank RCCL communicator
    (rt_comm_unique_id / rt_comm_init: the library opens RCCL) gives the image of rt_render."""
    from raytracing_amd import api
    api.init(0)
    o = orc.Scene.builtin(2)
    g = api.DeviceScene(o.desc, keep=o)
    org, d = rayset.mixed(orc, n_each=3000)
    for type_, n_obj in ((1, o.desc.n_spheres), (2, o.desc.n_squares)):
        for obj in range(n_obj):
            a, b = o.primitive_intersect(type_, obj, org, d), g.primitive_intersect(type_, obj, org, d)
            assert a["hit"].sum() > 50
            assert a.tobytes() == b.tobytes(), (type_, obj)
    with pytest.raises(api.RtError):
        g.primitive_intersect(1, 99, org, d)
    w, h, spp = 100, 37, 3
    cam = orc.camera_default(w, h)
    want, st = g.render(cam, w, h, spp, seed=9, tile_w=16, tile_h=8)
    got, st0 = g.render_sharded(cam, w, h, spp, seed=9, tile_w=16, tile_h=8)      # no communicator: rank 0 of 1
    assert got.tobytes() == want.tobytes() and st0.rays == st.rays
    api.comm_init(api.comm_unique_id(), 0, 1)
    try:
        assert api.lib().rt_comm_world() == 1 and api.lib().rt_comm_rank() == 0
        got, st1 = g.render_sharded(cam, w, h, spp, seed=9, tile_w=16, tile_h=8)
        assert got.tobytes() == want.tobytes() and st1.rays == st.rays
    finally:
        api.comm_destroy()
    g.close()


def test_native_frame_loop_costs_and_device_image():
    """The library's own multi-GPU frame loop at one rank on the headline frame (Cornell 1080p, 64 spp): rt_render_sharded and
    rt_render_sharded_device give the image of rt_render bit for bit; a second frame of the same size allocates nothing and
    costs little more than its kernels -- device-side un-tile, one copy of the final image (the phases are printed)."""
    import torch
    from raytracing_amd import api, workloads
    api.init(0)
    wl = workloads.make("c2")
    g = wl.upload()
    cam, w, h, spp = wl.camera(), wl.w, wl.h, wl.spp
    want, st = g.render(cam, w, h, spp, seed=1, tile_w=8, tile_h=8)
    got, _ = g.render_sharded(cam, w, h, spp, seed=1, tile_w=8, tile_h=8)          # first frame: buffers are made
    assert got.tobytes() == want.tobytes()
    pinned = torch.empty((h, w, 3), dtype=torch.float32, pin_memory=True)           # the caller's buffer is pinned: copied straight into it
    stats = api.Stats()
    c = cam if isinstance(cam, api.Camera) else api.copy_struct(api.Camera, cam)
    import ctypes as C
    for _ in range(2):
        api.check(api.lib().rt_render_sharded(g.h, C.byref(c), w, h, spp, 1, 8, 8, pinned.data_ptr(), C.byref(stats)))
    t = api.sharded_last_times()
    assert pinned.numpy().tobytes() == want.tobytes() and stats.rays == st.rays
    mb = w * h * 12 / 1e6
    print(f"rt_render_sharded, world 1, {w}x{h} {spp} spp into pinned memory: render {t.render_ms:.3f} ms, gather {t.gather_ms:.3f}, un-tile {t.untile_ms:.3f}, "
          f"copy {t.copy_ms:.3f} ms ({mb / t.copy_ms:.1f} GB/s), host copy {t.host_copy_ms:.3f}, whole call {t.total_ms:.3f} ms")
    assert t.host_copy_ms == 0.0 and mb / t.copy_ms >= 20.0, "the image leaves the device in one copy at the link's rate"
    assert t.total_ms <= t.render_ms + 1.5 + t.copy_ms, "fixed cost of a frame beyond its kernels and the copy"
    got2, _ = g.render_sharded(cam, w, h, spp, seed=1, tile_w=8, tile_h=8)          # pageable destination: through the pinned block
    t2 = api.sharded_last_times()
    print(f"... into pageable memory: copy {t2.copy_ms:.3f} + host copy {t2.host_copy_ms:.3f} ms, whole call {t2.total_ms:.3f} ms")
    assert got2.tobytes() == want.tobytes()
    d_img = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    g.render_sharded_device(cam, w, h, spp, 1, 8, 8, d_img.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d_img.cpu().numpy().tobytes() == want.tobytes()
    g.close()


def test_launcher_refuses_more_ranks_than_gpus(tmp_path):
    """rt_main --gpus N with fewer visible GPUs ends with an error instead of leaving ranks waiting for one another."""
    from raytracing_amd import api
    n = api.device_count()
    r = subprocess.run([os.path.join(HOST_DIR, "rt_main"), "--w", "64", "--h", "48", "--spp", "2", "--gpus", str(n + 1), "--out", str(tmp_path / "x.ppm"),
                        "--off", host.STAR_OFF], capture_output=True, timeout=120)
    assert r.returncode != 0 and b"HIP device" in r.stderr


def test_headless_plus_and_r_keys(tmp_path):
    """rt_main --keys +r : '+' selects scene 3, 'r' renders it to a P3 file; same bytes as the oracle's render
    of that scene under the counter RNG (up to the rare pixel whose colour sits on a quantisation step)."""
    out = tmp_path / "rendu.ppm"
    w, h, spp = 96, 80, 3
    subprocess.run([os.path.join(HOST_DIR, "rt_main"), "--w", str(w), "--h", str(h), "--spp", str(spp), "--seed", "1", "--scene", "2", "--keys", "+r",
                    "--out", str(out), "--p6", str(tmp_path / "rendu_p6.ppm"), "--off", host.STAR_OFF], check=True, capture_output=True,
                   timeout=300)
    toks = open(out).read().split()
    assert toks[:4] == ["P3", str(w), str(h), "255"]
    got = np.array(toks[4:], np.uint8).reshape(h, w, 3)
    raw = open(tmp_path / "rendu_p6.ppm", "rb").read()
    head = f"P6\n{w} {h}\n255\n".encode()
    assert raw.startswith(head) and raw[len(head):] == got.tobytes()  # the binary file holds the same pixels
    o = orc.Scene.builtin(3)
    img, _ = o.render(orc.camera_default(w, h), w, h, spp, orc.RT_RNG_COUNTER, seed=1, threads=4)
    want = orc.quantize(img)
    assert (got != want).mean() < 1e-3 and np.abs(got.astype(int) - want.astype(int)).max() <= 1


def test_scene_refit_meshes_after_a_move(tmp_path):
    """Scene::refitMeshes(): the mesh is translated after the upload; the refitted scene answers like one set up with the
    moved mesh (api_check compares both in one process; a 1 058-triangle mesh, so a BVH is refitted)."""
    from raytracing_amd import workloads
    off = workloads.dome_off(24, directory=str(tmp_path))
    org, d = rayset.primary_grid(orc, 24, 20, seed=5)
    o2, d2 = rayset.interior_rays(1500, seed=45)
    org, d = np.concatenate([org, o2]), np.concatenate([d, d2])
    rays = tmp_path / "rays.txt"
    with open(rays, "w") as f:
        for i in range(len(org)):
            f.write(" ".join(repr(float(x)) for x in (*org[i], *d[i])) + " 0 0\n")
    out = subprocess.run([os.path.join(HOST_DIR, "api_check"), "3", off, str(rays), "refit"], check=True,
                         capture_output=True, text=True, timeout=300).stdout
    last = [ln for ln in out.splitlines() if ln.startswith("REFIT")]
    assert len(last) == 1, out[-500:]
    w = last[0].split()
    assert w[1] == "done" and int(w[3]) == len(org) and int(w[5]) >= 20 and int(w[7]) == 0, last[0]
