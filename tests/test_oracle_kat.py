"""Pins the CPU oracle against the one known answer the reference ships.

R/rendu.ppm is the reference's own 480x480, 50 spp render of scene 3 (Cornell box + star mesh,
setup_cornell_box_meshes) under the never-seeded glibc rand().  tests/golden/
rendu_scene3_480x480_50spp.npz holds its decoded RGB bytes (tests/golden/make_golden.py made
it); the oracle must reproduce all 230 400 pixels byte for byte.  The case exercises the camera
unprojection, the root-box slab test, Square and Mesh intersection, the quad-light Phong loop
with its 16 shadow rays, the sequential rand() order and the PPM quantisation.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import orc


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "rendu_scene3_480x480_50spp.npz"))["rgb"]


def test_golden_fixture_is_the_reference_file(golden_dir):
    g = _golden(golden_dir)
    assert g.shape == (480, 480, 3) and g.dtype == np.uint8
    assert hashlib.sha256(g.tobytes()).hexdigest() == \
        "27a08d12327e28b9eda5299467bc237e0aae156b94913483421e6e17b3c930a8"


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


@pytest.mark.slow
def test_kat0_full_image(golden_dir):
    """All 480 rows, about 80 s on one core."""
    g = _golden(golden_dir)
    s = orc.Scene.builtin(3)
    rgb, st = s.render(orc.camera_default(480, 480), 480, 480, 50, orc.RT_RNG_LIBC, 1)
    q = orc.quantize(rgb)
    assert int((q != g).any(axis=2).sum()) == 0
    assert st.samples == 480 * 480 * 50
