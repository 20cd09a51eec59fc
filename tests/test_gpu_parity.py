"""GPU parity: the HIP path (through the C-ABI of include/rtamd.h) against the CPU oracle on the
same seeded inputs.  Bit-exact for hit / type / object / triangle index, t, position and normal;
colours within the stated tolerance (they differ only through powf).  Run with -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import orc
from raytracing_amd import api
from tests import rayset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    api.init(0)
    yield


@pytest.fixture(scope="module", params=[0, 1, 2, 3])
def scene_pair(request):
    o = orc.Scene.builtin(request.param)
    g = api.DeviceScene(o.desc, keep=o)
    yield request.param, o, g
    g.close()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_hits_equal(a, b, what):
    for f in ("hit", "type", "object", "tri"):
        bad = np.nonzero(a[f] != b[f])[0]
        assert bad.size == 0, f"{what}: {f} differs on {bad.size} rays, first {bad[:5]}: {a[f][bad[:5]]} vs {b[f][bad[:5]]}"
    for f in ("t", "u", "v", "p", "n"):
        bad = np.nonzero((_bits(a[f]) != _bits(b[f])).reshape(len(a), -1).any(axis=1))[0]
        assert bad.size == 0, f"{what}: {f} bits differ on {bad.size} rays, first {bad[:5]}"


def test_generate_rays_bit_exact():
    rng = np.random.default_rng(3)
    u = rng.random(20000).astype(np.float32)
    v = rng.random(20000).astype(np.float32)
    u[:4] = [0.0, 1.0, 0.5, 1.0]
    v[:4] = [0.0, 1.0, 0.5, 0.0]
    for w, h in ((480, 480), (1920, 1080)):
        cam = orc.camera_default(w, h)
        oo, od = orc.generate_rays(cam, u, v)
        go, gd = api.generate_rays(cam, u, v)
        assert np.array_equal(_bits(oo), _bits(go)) and np.array_equal(_bits(od), _bits(gd))
    # a rotated, shifted camera: general (dense) matrices through the cofactor inverse
    q = np.array([0.1, -0.2, 0.3, 0.9], np.float32)
    q /= np.float32(np.linalg.norm(q))
    cam = orc.camera_build(45.0, 4.1, 10000.0, 640, 360, 0.3, -0.2, -3.1, 3.0, q)
    oo, od = orc.generate_rays(cam, u, v)
    go, gd = api.generate_rays(cam, u, v)
    assert np.array_equal(_bits(oo), _bits(go)) and np.array_equal(_bits(od), _bits(gd))


def test_kdtree_matches_oracle(scene_pair):
    _, o, g = scene_pair
    on, orf, ob = o.kdtree()
    gn, grf, gb = g.kdtree()
    assert np.array_equal(on, gn) and np.array_equal(orf, grf) and np.array_equal(_bits(ob), _bits(gb))


def test_intersect_bit_exact(scene_pair):
    sid, o, g = scene_pair
    org, d = rayset.mixed(orc)
    a = o.intersect(org, d)
    b = g.intersect(org, d)
    assert a["hit"].sum() > 1000
    assert_hits_equal(a, b, f"scene {sid}")


def test_mesh_intersect_tindex():
    o = orc.Scene.builtin(3)
    g = api.DeviceScene(o.desc, keep=o)
    org, d = rayset.interior_rays(20000, seed=23)
    tgt = np.random.default_rng(5).uniform(-0.6, 0.6, (20000, 3)).astype(np.float32) + np.float32([0, -1.3, 0])
    d = orc.normalize_dirs((tgt - org).astype(np.float32))
    a = o.mesh_intersect(0, org, d)
    b = g.mesh_intersect(0, org, d)
    assert a["hit"].sum() > 2000 and len(np.unique(a["tri"][a["hit"] == 1])) == 24
    assert_hits_equal(a, b, "star mesh")
    g.close()


def test_empty_batch_and_errors(scene_pair):
    _, _, g = scene_pair
    assert len(g.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))) == 0
    with pytest.raises(api.RtError):
        api.check(api.lib().rt_scene_destroy(987654321))
    bad = orc.Scene.builtin(2)
    d = api.copy_struct(api.SceneDesc, bad.desc)
    d.n_spheres = 1
    sph = (api.Sphere * 1)()
    sph[0].material = 99
    d.spheres = sph
    with pytest.raises(api.RtError):
        api.DeviceScene(d)


# colours: identical arithmetic except powf (glibc powf on the CPU, double-precision power rounded
# once on the GPU) -> a unit-in-the-last-place effect on the specular term
COLOR_ATOL = 2e-6


def test_trace_per_sample_colors(scene_pair):
    sid, o, g = scene_pair
    w, h, spp = 64, 64, 4
    rng = np.random.default_rng(29)
    ys, xs = np.mgrid[0:h, 0:w]
    pix = np.repeat((xs + ys * w).ravel().astype(np.uint32), spp)
    smp = np.tile(np.arange(spp, dtype=np.uint32), w * h)
    u = ((np.repeat(xs.ravel(), spp) + rng.random(pix.size)) / w).astype(np.float32)
    v = ((np.repeat(ys.ravel(), spp) + rng.random(pix.size)) / h).astype(np.float32)
    org, d = orc.generate_rays(orc.camera_default(w, h), u, v)
    d = orc.normalize_dirs(d)  # Ray(pos, dir) normalises the already-unit direction once more (R/main.cpp:190)
    oc, ost = o.trace(org, d, pix, smp, seed=1, k0=2)
    gc, gst = g.trace(org, d, pix, smp, seed=1, k0=2)
    assert (ost.rays, ost.samples, ost.nodes) == (gst.rays, gst.samples, gst.nodes)
    assert np.isfinite(gc).all()
    assert np.abs(oc - gc).max() <= COLOR_ATOL, np.abs(oc - gc).max()


def _moved_cornell(scale, shift):
    """The built-in Cornell box (scene 2) scaled about the origin and translated: descriptor arrays of its own."""
    base = orc.Scene.builtin(2)
    d = api.copy_struct(api.SceneDesc, base.desc)
    shift = np.asarray(shift, np.float32)
    mv = lambda p: (np.asarray(list(p), np.float32) * np.float32(scale) + shift).astype(np.float32)
    spheres = (api.Sphere * d.n_spheres)()
    for i in range(d.n_spheres):
        spheres[i].center[:] = mv(d.spheres[i].center)
        spheres[i].radius = float(np.float32(d.spheres[i].radius) * np.float32(scale))
        spheres[i].material = d.spheres[i].material
    squares = (api.Square * d.n_squares)()
    for i in range(d.n_squares):
        for v in range(4):
            squares[i].v[v][:] = mv(d.squares[i].v[v])
        squares[i].material = d.squares[i].material
    lights = (api.Light * d.n_lights)()
    for i in range(d.n_lights):
        lights[i].type = d.lights[i].type
        lights[i].color[:] = list(d.lights[i].color)
        for name in ("pos", "p0", "p1", "p2"):
            getattr(lights[i], name)[:] = mv(getattr(d.lights[i], name))
    d.spheres, d.squares, d.lights = spheres, squares, lights
    return d, [base, spheres, squares, lights]


@pytest.mark.parametrize("scale,shift", [(137.0, (310.0, -220.0, 170.0)), (0.01, (0.0, 0.0, 0.0)), (1.0, (-2000.0, 1500.0, 900.0))])
def test_shadow_rules_hold_at_other_scales(scale, shift):
    """The rules that decide shadow rays without the objects' own arithmetic (light / square table, sphere certificates) carry
    margins derived from the magnitudes of the scene: a Cornell box 137 times as large and far from the origin, one a hundred
    times smaller, and one 2 500 units away must still agree with the oracle ray for ray (equal ray and node counts) and colour
    for colour."""
    d, keep = _moved_cornell(scale, shift)
    o = orc.Scene.from_desc(api.copy_struct(orc.SceneDesc, d), keep=keep)
    g = api.DeviceScene(d, keep=keep)
    w, h, spp = 48, 40, 4
    rng = np.random.default_rng(31)
    ys, xs = np.mgrid[0:h, 0:w]
    pix = np.repeat((xs + ys * w).ravel().astype(np.uint32), spp)
    smp = np.tile(np.arange(spp, dtype=np.uint32), w * h)
    u = ((np.repeat(xs.ravel(), spp) + rng.random(pix.size)) / w).astype(np.float32)
    v = ((np.repeat(ys.ravel(), spp) + rng.random(pix.size)) / h).astype(np.float32)
    org, dirs = orc.generate_rays(orc.camera_default(w, h), u, v)
    org = (org * np.float32(scale) + np.asarray(shift, np.float32)).astype(np.float32)
    dirs = orc.normalize_dirs(dirs)
    oc, ost = o.trace(org, dirs, pix, smp, seed=1, k0=2)
    gc, gst = g.trace(org, dirs, pix, smp, seed=1, k0=2)
    assert ost.nodes > 2 * pix.size // 3, "the rays must reach the box"
    assert (ost.rays, ost.samples, ost.nodes) == (gst.rays, gst.samples, gst.nodes)
    assert np.abs(oc - gc).max() <= COLOR_ATOL, np.abs(oc - gc).max()
    a, b = o.intersect(org, dirs), g.intersect(org, dirs)
    assert a.tobytes() == b.tobytes()
    g.close()


def psnr(a, b):
    mse = np.mean((np.clip(a, 0, 1).astype(np.float64) - np.clip(b, 0, 1)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


def test_render_image_psnr(scene_pair):
    sid, o, g = scene_pair
    w, h, spp = 96, 54, 6
    cam = orc.camera_default(w, h)
    oi, ost = o.render(cam, w, h, spp, orc.RT_RNG_COUNTER, seed=5, threads=4)
    gi, gst = g.render(cam, w, h, spp, seed=5, tile_w=16, tile_h=8)
    assert (ost.rays, ost.samples, ost.nodes) == (gst.rays, gst.samples, gst.nodes)
    assert psnr(oi, gi) >= 40.0  # the acceptance bar; in practice the images agree to ~1e-6
    assert np.abs(oi - gi).max() <= COLOR_ATOL
    assert np.array_equal(orc.quantize(oi), orc.quantize(gi)) or (orc.quantize(oi) != orc.quantize(gi)).mean() < 1e-3


def test_tile_sharding_is_exact():
    """Rendering the tiles of 2 or 3 ranks separately and scattering them gives the 1-rank image bit for bit
    (ragged image size: 100 x 37 is not a multiple of the tile)."""
    o = orc.Scene.builtin(2)
    g = api.DeviceScene(o.desc, keep=o)
    w, h, spp, tw, th = 100, 37, 3, 16, 8
    cam = orc.camera_default(w, h)
    ref, _ = g.render(cam, w, h, spp, seed=9, tile_w=tw, tile_h=th)
    for world in (2, 3):
        per = api.tiles_per_rank(w, h, tw, th, 0, world)
        buf = np.zeros((world, per * th * tw * 3), np.float32)
        rays = 0
        for r in range(world):
            tiles, st = g.render_tiles(cam, w, h, spp, 9, tw, th, r, world)
            buf[r, :tiles.size] = tiles.ravel()
            rays += st.rays
        img = api.tiles_to_image(buf, w, h, tw, th, world)
        assert np.array_equal(_bits(img), _bits(ref))
    g.close()


def test_progressive_passes_give_the_one_shot_floats():
    """Samples split over passes (packed-lane passes of 20 and 10, a wave-per-pixel pass of 40) continue the
    float sum in sample order: the result is the one-shot render of 70 spp bit for bit, on a ragged image."""
    import torch
    o = orc.Scene.builtin(2)
    g = api.DeviceScene(o.desc, keep=o)
    w, h, tw, th, total = 90, 50, 16, 16, 70
    cam = orc.camera_default(w, h)
    ref, _ = g.render_tiles(cam, w, h, total, 11, tw, th, 0, 1)
    dev = torch.device("cuda", 0)
    buf = torch.full((ref.size,), 123.0, dtype=torch.float32, device=dev)  # stale contents must not matter
    stream = torch.cuda.current_stream(dev).cuda_stream
    s0 = 0
    for n in (20, 40, 10):
        g.render_pass_device(cam, w, h, n, s0, total, 11, tw, th, 0, 1, buf.data_ptr(), 0, stream)
        s0 += n
    torch.cuda.synchronize()
    assert np.array_equal(_bits(buf.cpu().numpy()), _bits(ref.ravel()))
    with pytest.raises(api.RtError):
        g.render_pass_device(cam, w, h, 10, 65, total, 11, tw, th, 0, 1, buf.data_ptr(), 0, stream)
    g.close()


def test_packed_lanes_match_the_oracle_for_every_small_spp():
    """spp <= 32 puts 64 / spp pixels into one wave; every such grouping (including the ones that leave
    lanes idle, and a tile whose pixel count is not a multiple of the group) matches the oracle image."""
    o = orc.Scene.builtin(2)
    g = api.DeviceScene(o.desc, keep=o)
    w, h = 37, 21
    cam = orc.camera_default(w, h)
    for spp in (1, 2, 3, 5, 7, 8, 13, 21, 32, 33, 50, 63, 64, 65, 100, 130):  # above 32: cut into 64s, a 32 and a rest
        gi, gst = g.render(cam, w, h, spp, seed=spp, tile_w=5, tile_h=3)
        oi, ost = o.render(cam, w, h, spp, orc.RT_RNG_COUNTER, spp)
        assert gst.rays == ost.rays and gst.samples == ost.samples == w * h * spp
        assert np.abs(gi - oi).max() <= COLOR_ATOL, spp
    g.close()


def test_device_untile_and_quantise_matches_host_path():
    """rt_tiles_to_rgb8_device on the gathered buffers of 3 ranks = rt_tiles_to_image + the reference's quantisation."""
    import torch
    o = orc.Scene.builtin(2)
    g = api.DeviceScene(o.desc, keep=o)
    w, h, spp, tw, th, world = 100, 37, 4, 16, 8, 3
    cam = orc.camera_default(w, h)
    per = api.tiles_per_rank(w, h, tw, th, 0, world)
    buf = np.zeros((world, per * th * tw * 3), np.float32)
    for r in range(world):
        tiles, _ = g.render_tiles(cam, w, h, spp, 9, tw, th, r, world)
        buf[r, :tiles.size] = tiles.ravel()
    buf[0, :3] = [2.5, -0.25, np.nan]  # above 1, negative, NaN: whatever the CPU expression gives
    want = orc.quantize(api.tiles_to_image(buf, w, h, tw, th, world))
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(buf).to(dev)
    d_out = torch.zeros(h * w * 3, dtype=torch.uint8, device=dev)
    api.tiles_to_rgb8_device(d_in.data_ptr(), w, h, tw, th, world, d_out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().reshape(h, w, 3), want)
    g.close()


def test_tile_launch_order_does_not_change_the_image(tmp_path):
    """The deepest-path-first launch order of the tiles only decides when a tile is rendered: a second
    process with RTAMD_TILE_ORDER=0 (index order) produces the same floats, sharded or not."""
    import subprocess
    import sys
    w, h, spp, tw, th = 200, 120, 3, 16, 16
    out = tmp_path / "plain.npz"
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from raytracing_amd import api, host\n"
        "api.init(0)\n"
        "g = host.HostScene.builtin(2).upload()\n"
        f"cam = host.camera_default({w}, {h})\n"
        f"full, _ = g.render(cam, {w}, {h}, {spp}, seed=3, tile_w={tw}, tile_h={th})\n"
        f"part, _ = g.render_tiles(cam, {w}, {h}, {spp}, 3, {tw}, {th}, 1, 3)\n"
        f"np.savez({str(out)!r}, full=full, part=part)\n")
    env = dict(os.environ, RTAMD_TILE_ORDER="0")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=100)
    from raytracing_amd import host
    g = host.HostScene.builtin(2).upload()
    cam = host.camera_default(w, h)
    full, _ = g.render(cam, w, h, spp, seed=3, tile_w=tw, tile_h=th)
    part, _ = g.render_tiles(cam, w, h, spp, 3, tw, th, 1, 3)
    ref = np.load(out)
    assert np.array_equal(_bits(full), _bits(ref["full"])) and np.array_equal(_bits(part), _bits(ref["part"]))
    assert full.mean() > 0.05
    g.close()


@pytest.mark.parametrize("scene", [3, 2])
def test_matches_reference_renders_statistically(golden_dir, scene):
    """The GPU render (50 spp, counter RNG) against the reference's own images of scene 3 (rendu.ppm) and scene 2
    (the report's Phase4/renduSphereVerre2.png): different random numbers, same estimator, so the images agree up
    to Monte-Carlo noise -- on the walls, in the soft shadows, in the mirror and through the glass."""
    gold = np.load(os.path.join(golden_dir, f"rendu_scene{scene}_480x480_50spp.npz"))["rgb"].astype(np.float64) / 255
    o = orc.Scene.builtin(scene)
    g = api.DeviceScene(o.desc, keep=o)
    img, st = g.render(orc.camera_default(480, 480), 480, 480, 50, seed=1)
    assert st.samples == 480 * 480 * 50
    if scene == 3:  # 1 primary + 16 shadow rays per sample, minus the few samples whose primary ray slips between two walls
        assert 16.999 * st.samples < st.rays <= 17 * st.samples
    q = orc.quantize(img).astype(np.float64) / 255
    assert abs(q.mean() - gold.mean()) < 2e-3
    assert psnr(q, gold) > 30.0
    g.close()


# ---- scenes with more than 30 objects: the object kd-tree has internal nodes, KDTree::intersectNode's
# near / far rules (and its misses) must be reproduced exactly --------------------------------------------
from tests import scenegen  # noqa: E402


@pytest.fixture(scope="module", params=[(40, 24, 3, False), (70, 0, 5, True), (10, 60, 9, False)])
def kd_pair(request):
    ns, nq, seed, point = request.param
    d, keep = scenegen.random_objects(orc, ns, nq, seed, point_light=point)
    o = orc.Scene.from_desc(d, keep)
    g = api.DeviceScene(o.desc, keep=o)
    yield o, g
    g.close()


def test_kd_internal_nodes_tree_and_hits(kd_pair):
    o, g = kd_pair
    on, orf, ob = o.kdtree()
    gn, grf, gb = g.kdtree()
    assert (on[:, 0] == 0).sum() >= 1, "scene must produce internal kd nodes"
    assert np.array_equal(on, gn) and np.array_equal(orf, grf) and np.array_equal(_bits(ob), _bits(gb))
    org, d = rayset.mixed(orc, n_each=6000)
    a, b = o.intersect(org, d), g.intersect(org, d)
    assert a["hit"].sum() > 1000
    assert_hits_equal(a, b, "kd scene")


def test_kd_internal_nodes_trace(kd_pair):
    o, g = kd_pair
    w, h, spp = 48, 48, 2
    rng = np.random.default_rng(31)
    ys, xs = np.mgrid[0:h, 0:w]
    pix = np.repeat((xs + ys * w).ravel().astype(np.uint32), spp)
    smp = np.tile(np.arange(spp, dtype=np.uint32), w * h)
    u = ((np.repeat(xs.ravel(), spp) + rng.random(pix.size)) / w).astype(np.float32)
    v = ((np.repeat(ys.ravel(), spp) + rng.random(pix.size)) / h).astype(np.float32)
    org, d = orc.generate_rays(orc.camera_default(w, h), u, v)
    d = orc.normalize_dirs(d)
    oc, ost = o.trace(org, d, pix, smp, seed=2, k0=2)
    gc, gst = g.trace(org, d, pix, smp, seed=2, k0=2)
    assert (ost.rays, ost.samples, ost.nodes) == (gst.rays, gst.samples, gst.nodes)
    both = np.isfinite(oc) & np.isfinite(gc)
    assert (np.isfinite(oc) == np.isfinite(gc)).all()
    assert np.abs(oc[both] - gc[both]).max() <= 5e-6  # generic shininess goes through pow(): a few ulp more than the x^16 path
    img_o, so = o.render(orc.camera_default(40, 30), 40, 30, 2, orc.RT_RNG_COUNTER, seed=4, threads=4)
    img_g, sg = g.render(orc.camera_default(40, 30), 40, 30, 2, seed=4, tile_w=8, tile_h=8)
    assert so.rays == sg.rays
    ok = np.isfinite(img_o) & np.isfinite(img_g)
    assert np.abs(img_o[ok] - img_g[ok]).max() <= 5e-6


# ---- large meshes: the per-mesh triangle BVH must give Mesh::intersect's answer ---------------------------
def _mesh_scene(n_grid, seed=1, spheres=2, squares=5):
    mesh = scenegen.grid_mesh(n_grid, seed=seed)
    d, keep = scenegen.random_objects(orc, spheres, squares, seed, with_mesh=mesh)
    o = orc.Scene.from_desc(d, keep)
    return o, api.DeviceScene(o.desc, keep=o), mesh


def _grazing_report(a, b):
    """Indices where the BVH path and the brute-force oracle disagree on hit / triangle."""
    return np.nonzero((a["hit"] != b["hit"]) | (a["tri"] != b["tri"]) | (a["object"] != b["object"]) | (a["type"] != b["type"]))[0]


def test_bvh_mesh_intersect_matches_brute_force():
    o, g, (pos, nrm, tri) = _mesh_scene(72)       # 2 * 71^2 = 10 082 triangles
    rng = np.random.default_rng(41)
    n = 30000
    org = rng.uniform(-1.9, 1.9, (n, 3)).astype(np.float32)
    org[:, 1] = rng.uniform(-0.9, 1.9, n).astype(np.float32)            # mostly above the height field
    tgt = pos[rng.integers(0, len(pos), n)] + rng.normal(scale=0.05, size=(n, 3)).astype(np.float32)
    d = (tgt - org).astype(np.float32)
    d[: n // 10] = rng.normal(size=(n // 10, 3)).astype(np.float32)     # and some that mostly miss
    d = orc.normalize_dirs(d)
    a = o.mesh_intersect(0, org, d)
    b = g.mesh_intersect(0, org, d)
    assert a["hit"].sum() > n // 2
    bad = _grazing_report(a, b)
    # the contract allows a reported list of grazing hits; the padded boxes make it empty here
    assert bad.size == 0, f"{bad.size} rays differ, first {bad[:5]}: ref tri {a['tri'][bad[:5]]} t {a['t'][bad[:5]]} vs {b['tri'][bad[:5]]} t {b['t'][bad[:5]]}"
    assert_hits_equal(a, b, "bvh mesh")
    g.close()


def test_bvh_scene_intersect_and_trace():
    o, g, _ = _mesh_scene(48, seed=2)             # 4 418 triangles + 2 spheres + 5 squares
    org, d = rayset.mixed(orc, n_each=1500)
    a, b = o.intersect(org, d), g.intersect(org, d)
    assert (a["type"][a["hit"] == 1] == 0).sum() > 200, "rays must reach the mesh"
    assert_hits_equal(a, b, "bvh scene")
    w, h, spp = 40, 30, 1
    img_o, so = o.render(orc.camera_default(w, h), w, h, spp, orc.RT_RNG_COUNTER, seed=6, threads=4)
    img_g, sg = g.render(orc.camera_default(w, h), w, h, spp, seed=6, tile_w=8, tile_h=8)
    assert (so.rays, so.nodes) == (sg.rays, sg.nodes)
    ok = np.isfinite(img_o) & np.isfinite(img_g)
    assert np.abs(img_o[ok] - img_g[ok]).max() <= 5e-6
    g.close()


def test_bvh_100k_triangles_spot_check(tmp_path):
    """Full-size mesh (100 352 triangles, loaded from an OFF file by the host Mesh class): 1 500 rays against the
    brute-force oracle (which needs ~1.5e8 triangle tests for them)."""
    from raytracing_amd import host
    from tools import make_mesh_off
    off = str(tmp_path / "grid225.off")
    p, t = make_mesh_off.make(225)
    make_mesh_off.write_off(off, p, t)
    hs = host.HostScene.mesh_file(off, with_spheres=True)
    assert hs.desc.n_triangles == 100352
    o = orc.Scene.from_desc(api.copy_struct(orc.SceneDesc, hs.desc), keep=hs)
    g = hs.upload()
    org, d = rayset.primary_grid(orc, 40, 25, seed=3)
    o2, d2 = rayset.interior_rays(300, seed=5)
    verts = np.ctypeslib.as_array(hs.desc.positions, shape=(hs.desc.n_vertices, 3))
    rng = np.random.default_rng(8)
    o3 = rng.uniform(-1.8, 1.8, (700, 3)).astype(np.float32)
    o3[:, 1] = rng.uniform(-0.5, 1.8, 700).astype(np.float32)
    d3 = (verts[rng.integers(0, len(verts), 700)] + rng.normal(scale=0.01, size=(700, 3)).astype(np.float32) - o3).astype(np.float32)
    org, d = np.concatenate([org, o2, o3]), orc.normalize_dirs(np.concatenate([d, d2, d3]))
    a, b = o.intersect(org, d), g.intersect(org, d)
    assert (a["type"][a["hit"] == 1] == 0).sum() > 400
    bad = _grazing_report(a, b)
    assert bad.size == 0, f"{bad.size} grazing mismatches: {bad[:8]}"
    assert_hits_equal(a, b, "100k mesh")
    g.close()


def test_shortened_divide_and_sqrt_equal_the_ieee_operators():
    """2^28 random vectors (exponents 2^-44 .. 2^44, zeros mixed in): the fast normalisation path must
    return the bits of sqrt and `/` everywhere, inside and outside the range it accepts."""
    assert api.selftest_arith(1 << 28, seed=7) == 0


def test_progressive_renderer_matches_the_one_shot_image():
    """parallel_render.ProgressiveRenderer on one GPU: sums stay in the tile buffer between passes, display()
    shows sum / samples so far, and the last display is the one-shot image bit for bit."""
    import torch
    from raytracing_amd import parallel_render
    o = orc.Scene.builtin(2)
    g = api.DeviceScene(o.desc, keep=o)
    w, h, spp, tw, th = 80, 45, 40, 16, 16
    cam = orc.camera_default(w, h)
    dev = torch.device("cuda", 0)
    want, _ = g.render(cam, w, h, spp, seed=4, tile_w=tw, tile_h=th)
    first, _ = g.render(cam, w, h, 8, seed=4, tile_w=tw, tile_h=th)      # the first 8 samples alone, averaged
    r = parallel_render.ProgressiveRenderer(w, h, tw, th, dev, parallel_render.device_pass_fn(g, cam, w, h, 4, tw, th), spp)
    r.add_pass(8)
    img = r.display().cpu().numpy()
    assert np.abs(img - first).max() <= 1e-6       # sum / 8 here, (sum / 8) there: the same division
    r.add_pass(12)
    r.add_pass(20)
    torch.cuda.synchronize()
    assert r.display().cpu().numpy().tobytes() == want.tobytes()
    g.close()


def test_full_size_frame_properties():
    """The benchmark frame itself (Cornell box, 1920 x 1080, 64 spp), through properties that need no oracle run of
    that size: the 8-rank split reproduces the 1-rank image and ray count bit for bit, four passes of 16 samples
    reproduce the one-shot floats, the image is finite and non-negative, and six full pixel rows at 2 spp equal the
    oracle's colours (same counter RNG, so the very same paths)."""
    import torch
    o = orc.Scene.builtin(2)
    g = api.DeviceScene(o.desc, keep=o)
    w, h, spp, tw, th = 1920, 1080, 64, 16, 16
    cam = orc.camera_default(w, h)
    ref, st = g.render(cam, w, h, spp, seed=1, tile_w=tw, tile_h=th)
    assert st.samples == w * h * spp
    assert np.isfinite(ref).all() and ref.min() >= 0.0   # no upper bound: the reference never clamps (R/src/Scene.h:713)
    world = 8
    per = api.tiles_per_rank(w, h, tw, th, 0, world)
    buf = np.zeros((world, per * th * tw * 3), np.float32)
    rays = nodes = 0
    for r in range(world):
        tiles, s = g.render_tiles(cam, w, h, spp, 1, tw, th, r, world)
        buf[r, :tiles.size] = tiles.ravel()
        rays += s.rays
        nodes += s.nodes
    assert (rays, nodes) == (st.rays, st.nodes)
    assert np.array_equal(_bits(api.tiles_to_image(buf, w, h, tw, th, world)), _bits(ref))
    dev = torch.device("cuda", 0)
    n1 = api.tiles_per_rank(w, h, tw, th, 0, 1)
    acc = torch.zeros(n1 * th * tw * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for s0 in range(0, spp, 16):
        g.render_pass_device(cam, w, h, 16, s0, spp, 1, tw, th, 0, 1, acc.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    img = api.tiles_to_image(acc.cpu().numpy().reshape(1, -1), w, h, tw, th, 1)
    assert np.array_equal(_bits(img), _bits(ref))
    # the oracle on 6 full rows of the frame at 2 spp: the same counter RNG, hence the very same paths
    sub_g, _ = g.render(cam, w, h, 2, seed=1, tile_w=tw, tile_h=th)
    for y in (0, 217, 540, 700, 863, 1079):
        row_o, _ = o.render(cam, w, h, 2, orc.RT_RNG_COUNTER, seed=1, y0=y, y1=y + 1, threads=2)
        assert np.abs(sub_g[y] - row_o[y]).max() <= 5e-6, y
    g.close()
