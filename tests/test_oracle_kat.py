"""Pins the CPU oracle against the two known answers the reference ships.

KAT-0: R/rendu.ppm is the reference's own 480x480, 50 spp render of scene 3 (Cornell box + star mesh,
setup_cornell_box_meshes) under the never-seeded glibc rand().  It exercises the camera unprojection, the
root-box slab test, Square and Mesh intersection, the quad-light Phong loop with its 16 shadow rays, the
sequential rand() order and the PPM quantisation.
KAT-1: R/Compte-rendu/Images/Phase4/renduSphereVerre2.png is the report's final render of scene 2 (Cornell
box with the mirror and the glass sphere, setup_cornell_box), same flow, saved as PNG.  It adds Sphere
intersection, the mirror and glass recursion with its Fresnel terms and total internal reflection, and the
rand() order across recursion levels.
tests/golden/rendu_scene{3,2}_480x480_50spp.npz hold the decoded RGB bytes (tests/golden/make_golden.py made
them); the oracle must reproduce all 230 400 pixels of each byte for byte.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import orc


def _golden(golden_dir, scene=3):
    return np.load(os.path.join(golden_dir, f"rendu_scene{scene}_480x480_50spp.npz"))["rgb"]


SHA256 = {3: "27a08d12327e28b9eda5299467bc237e0aae156b94913483421e6e17b3c930a8",
          2: "200b8dd54f217c4ddba7c759841c2ff4af675cff0bd3b1b0741d5e844651975f"}


@pytest.mark.parametrize("scene", [3, 2])
def test_golden_fixture_is_the_reference_file(golden_dir, scene):
    g = _golden(golden_dir, scene)
    assert g.shape == (480, 480, 3) and g.dtype == np.uint8
    assert hashlib.sha256(g.tobytes()).hexdigest() == SHA256[scene]


def test_libm_variants_agree_on_wall_angles():
    assert orc.lib().orc_selfcheck() == 0


def test_default_camera_matrices():
    cam = orc.camera_default(480, 480)
    p = np.array(cam.projection, np.float32)
    m = np.array(cam.modelview, np.float32)
    assert p[0] == np.float32(2.41421366) and p[5] == np.float32(2.41421366)
    assert p[10] == np.float32(-1.00082028) and p[14] == np.float32(-8.20336342) and p[11] == -1
    assert m[14] == np.float32(-3.1) + np.float32(-3.0)
    assert (m[[0, 5, 10, 15]] == 1).all()


def test_kat0_first_rows(golden_dir):
    """The first 24 rows (sequential rand() stream from the top of the image): fast smoke of KAT-0."""
    g = _golden(golden_dir)
    s = orc.Scene.builtin(3)
    rgb, st = s.render(orc.camera_default(480, 480), 480, 480, 50, orc.RT_RNG_LIBC, 1, 0, 24)
    assert (orc.quantize(rgb)[:24] == g[:24]).all()
    assert st.samples == 24 * 480 * 50 and st.rays == 17 * st.samples


def test_kat1_first_rows(golden_dir):
    """Scene 2, the first 24 rows (ceiling, walls, the light's highlight): fast smoke of KAT-1."""
    g = _golden(golden_dir, 2)
    s = orc.Scene.builtin(2)
    rgb, st = s.render(orc.camera_default(480, 480), 480, 480, 50, orc.RT_RNG_LIBC, 1, 0, 24)
    assert (orc.quantize(rgb)[:24] == g[:24]).all()


def _full_render(scene):
    """Worker of the full-image test (its own process: the libc rand() stream is process-wide state)."""
    s = orc.Scene.builtin(scene)
    rgb, st = s.render(orc.camera_default(480, 480), 480, 480, 50, orc.RT_RNG_LIBC, 1)
    return orc.quantize(rgb), int(st.samples), int(st.rays)


@pytest.mark.slow
def test_kat0_and_kat1_full_images(golden_dir):
    """All 480 rows of both images, rendered in two processes side by side: about 80 s (scene 3) and 150 s
    (scene 2) of one core each."""
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(2) as pool:
        results = pool.map(_full_render, [3, 2])
    for scene, (q, samples, rays) in zip([3, 2], results):
        g = _golden(golden_dir, scene)
        assert int((q != g).any(axis=2).sum()) == 0, f"scene {scene}"
        assert samples == 480 * 480 * 50
    assert results[0][2] == 17 * results[0][1]  # scene 3: every sample is one primary + 16 shadow rays
