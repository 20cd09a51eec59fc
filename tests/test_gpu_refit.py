"""rt_scene_refit: the vertices of a mesh move, its triangles stay.  The refitted scene must answer exactly like a scene
created from the moved vertices -- and like the brute-force oracle on them -- for both tree builders, pass the host
verifier, and render the same floats.  (The reference has nothing that moves: the check is against a fresh build.)
Run with -m gpu."""
import numpy as np
import pytest

from oracle import orc
from raytracing_amd import api, host
from tests import rayset, scenegen
from tests.test_gpu_parity import assert_hits_equal, _bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    api.init(0)
    yield


def _deformed(pos, nrm, seed, amount):
    """The mesh bent, stretched and roughened, and moved: new float32 positions and (re-normalised, tilted) normals."""
    rng = np.random.default_rng(seed)
    p = pos.astype(np.float64)
    q = p.copy()
    q[:, 1] = p[:, 1] * (1.0 + amount) + amount * 0.4 * np.sin(3.0 * p[:, 0]) * np.cos(2.0 * p[:, 2]) + 0.05 * amount
    q[:, 0] = p[:, 0] * (1.0 + 0.3 * amount) + 0.1 * amount
    q += rng.normal(scale=0.01 * amount, size=q.shape)
    n = nrm.astype(np.float64) + rng.normal(scale=0.2 * amount, size=nrm.shape)
    n /= np.linalg.norm(n, axis=1)[:, None]
    return q.astype(np.float32), n.astype(np.float32)


def _scene_with(mesh, seed, api_mod):
    d, keep = scenegen.random_objects(api_mod, 2, 5, seed, with_mesh=mesh)
    return d, keep


def _mesh_rays(pos, n, seed):
    rng = np.random.default_rng(seed)
    org = rng.uniform(-1.9, 1.9, (n, 3)).astype(np.float32)
    org[:, 1] = rng.uniform(-0.9, 1.9, n).astype(np.float32)
    tgt = pos[rng.integers(0, len(pos), n)] + rng.normal(scale=0.05, size=(n, 3)).astype(np.float32)
    d = (tgt - org).astype(np.float32)
    d[: n // 10] = rng.normal(size=(n // 10, 3)).astype(np.float32)
    return org, orc.normalize_dirs(d)


@pytest.mark.parametrize("flags", [0, api.BUILD_BVH_DEVICE])
@pytest.mark.parametrize("amount", [0.05, 1.0])   # a small deformation, and one that leaves the old boxes far behind
def test_refitted_scene_equals_a_fresh_one(flags, amount):
    pos, nrm, tri = scenegen.grid_mesh(72, seed=3)          # 10 082 triangles
    d0, keep0 = _scene_with((pos, nrm, tri), 5, orc)
    g = api.DeviceScene(d0, keep=keep0, flags=flags)
    pos1, nrm1 = _deformed(pos, nrm, 7, amount)
    ms = g.refit(pos1, nrm1)
    assert ms > 0.0
    info, _ = g.verify()                                    # boxes nested, every triangle inside its leaf's box, both tables
    assert info.bvh_meshes == 1
    d1, keep1 = _scene_with((pos1, nrm1, tri), 5, orc)
    o1 = orc.Scene.from_desc(d1, keep1)
    fresh = api.DeviceScene(d1, keep=keep1, flags=flags)
    org, dr = _mesh_rays(pos1, 20000, seed=11)
    want = o1.mesh_intersect(0, org, dr)                    # brute force over the moved triangles
    assert want["hit"].sum() > 4000
    assert_hits_equal(want, g.mesh_intersect(0, org, dr), "refitted, mesh")
    org, dr = rayset.mixed(orc, n_each=1000)
    a, b, c = o1.intersect(org, dr), g.intersect(org, dr), fresh.intersect(org, dr)
    assert_hits_equal(a, b, "refitted, scene")
    assert_hits_equal(a, c, "fresh, scene")
    # the root box follows the mesh (rt_scene_kdtree reports it) and the frames are the same floats
    cam = host.camera_default(96, 64)
    ia, sa = g.render(cam, 96, 64, 4, seed=2)
    ib, sb = fresh.render(cam, 96, 64, 4, seed=2)
    assert np.array_equal(_bits(ia), _bits(ib)) and (sa.rays, sa.nodes) == (sb.rays, sb.nodes)
    # and back: the first geometry again gives the first answers
    g.refit(pos, nrm)
    o0 = orc.Scene.from_desc(d0, keep0)
    org, dr = _mesh_rays(pos, 5000, seed=12)
    assert_hits_equal(o0.mesh_intersect(0, org, dr), g.mesh_intersect(0, org, dr), "refitted back")
    g.close(); fresh.close()


def test_refit_rejects_what_it_cannot_do():
    pos, nrm, tri = scenegen.grid_mesh(12, seed=1)
    d, keep = scenegen.random_objects(orc, 40, 24, 3, with_mesh=(pos, nrm, tri))   # 65 objects: kd-tree with internal nodes
    g = api.DeviceScene(d, keep=keep)
    with pytest.raises(api.RtError):
        g.refit(pos, nrm)
    g.close()
    d, keep = _scene_with((pos, nrm, tri), 5, orc)
    g = api.DeviceScene(d, keep=keep)
    with pytest.raises(api.RtError):
        g.refit(pos[:-1], nrm[:-1])                           # another vertex count
    g.refit(pos, nrm)                                         # small mesh without a BVH: records and bounds only
    org, dr = _mesh_rays(pos, 2000, seed=3)
    o = orc.Scene.from_desc(d, keep)
    assert_hits_equal(o.mesh_intersect(0, org, dr), g.mesh_intersect(0, org, dr), "refit without a BVH")
    g.close()


@pytest.mark.parametrize("flags", [0, api.BUILD_BVH_DEVICE])
def test_refit_of_two_meshes(flags):
    """Two meshes with a BVH in one scene (their nodes share one table, each tree has its own pad), moved differently."""
    m1 = scenegen.grid_mesh(20, seed=3)                                   # 722 triangles
    m2 = scenegen.grid_mesh(14, seed=4, size=0.5, center=(0.8, 0.2, -0.5))  # 338 triangles
    d0, keep0 = scenegen.random_objects(orc, 1, 3, 5, with_mesh=m1, second_mesh=m2)
    g = api.DeviceScene(d0, keep=keep0, flags=flags)
    assert g.verify()[0].bvh_meshes == 2
    p1, n1 = _deformed(m1[0], m1[1], 21, 0.6)
    p2, n2 = _deformed(m2[0], m2[1], 22, 0.3)
    p2 = (p2 + np.float32([-0.4, 0.3, 0.2])).astype(np.float32)
    g.refit(np.concatenate([p1, p2]), np.concatenate([n1, n2]))
    g.verify()
    d1, keep1 = scenegen.random_objects(orc, 1, 3, 5, with_mesh=(p1, n1, m1[2]), second_mesh=(p2, n2, m2[2]))
    o1 = orc.Scene.from_desc(d1, keep1)
    org, dr = _mesh_rays(np.concatenate([p1, p2]), 12000, seed=13)
    for m in (0, 1):
        want = o1.mesh_intersect(m, org, dr)
        assert want["hit"].sum() > 500
        assert_hits_equal(want, g.mesh_intersect(m, org, dr), f"refitted mesh {m}")
    assert_hits_equal(o1.intersect(org, dr), g.intersect(org, dr), "refitted, two meshes")
    g.close()


def test_treelet_order_changes_the_table_not_the_answers():
    """The same two-mesh scene with its wide nodes in collapse order and in treelet order: both verify, hit records
    and frames are the same bits, and a refit keeps the order it was built with."""
    m1 = scenegen.grid_mesh(40, seed=3)
    m2 = scenegen.grid_mesh(14, seed=4, size=0.5, center=(0.8, 0.2, -0.5))
    d0, keep0 = scenegen.random_objects(orc, 1, 3, 5, with_mesh=m1, second_mesh=m2)
    plain = api.DeviceScene(d0, keep=keep0, flags=api.BUILD_BVH_DEVICE)
    api.set_option("treelet_order", 1)
    try:
        g = api.DeviceScene(d0, keep=keep0, flags=api.BUILD_BVH_DEVICE)
    finally:
        api.set_option("treelet_order", 0)
    assert g.verify()[0].bvh_nodes == plain.verify()[0].bvh_nodes
    org, dr = _mesh_rays(np.concatenate([m1[0], m2[0]]), 12000, seed=5)
    assert_hits_equal(plain.intersect(org, dr), g.intersect(org, dr), "treelet order")
    cam = host.camera_default(64, 48)
    ia, sa = plain.render(cam, 64, 48, 4, seed=2)
    ib, sb = g.render(cam, 64, 48, 4, seed=2)
    assert np.array_equal(_bits(ia), _bits(ib)) and (sa.rays, sa.nodes) == (sb.rays, sb.nodes)
    p1, n1 = _deformed(m1[0], m1[1], 21, 0.6)
    p2, n2 = _deformed(m2[0], m2[1], 22, 0.3)
    for s in (plain, g):
        s.refit(np.concatenate([p1, p2]), np.concatenate([n1, n2]))
        s.verify()
    d1, keep1 = scenegen.random_objects(orc, 1, 3, 5, with_mesh=(p1, n1, m1[2]), second_mesh=(p2, n2, m2[2]))
    o1 = orc.Scene.from_desc(d1, keep1)
    org, dr = _mesh_rays(np.concatenate([p1, p2]), 8000, seed=6)
    assert_hits_equal(o1.intersect(org, dr), g.intersect(org, dr), "treelet order, refitted")
    plain.close(); g.close()
