"""Committed oracle vectors (tests/golden/vectors_scene*.npz, made by tests/golden/make_vectors.py):
the CPU test freezes the oracle against them, the GPU test holds the HIP path to the same numbers."""
import os

import numpy as np
import pytest

from oracle import orc


def _load(golden_dir, sid):
    return np.load(os.path.join(golden_dir, f"vectors_scene{sid}.npz"))


@pytest.mark.parametrize("sid", [0, 1, 2, 3])
def test_oracle_reproduces_committed_vectors(golden_dir, sid):
    g = _load(golden_dir, sid)
    s = orc.Scene.builtin(sid)
    assert s.intersect(g["org"], g["dir"]).tobytes() == g["hits"].tobytes()
    rgb, st = s.trace(g["t_org"], g["t_dir"], g["t_pixel"], g["t_sample"], seed=1, k0=2)
    assert rgb.tobytes() == g["t_rgb"].tobytes() and [st.rays, st.samples, st.nodes] == list(g["t_counts"])
    img, ist = s.render(orc.camera_default(48, 27), 48, 27, 2, orc.RT_RNG_COUNTER, seed=3)
    assert img.tobytes() == g["img"].tobytes() and [ist.rays, ist.samples, ist.nodes] == list(g["img_counts"])


@pytest.mark.gpu
@pytest.mark.parametrize("sid", [0, 1, 2, 3])
def test_gpu_reproduces_committed_vectors(golden_dir, sid):
    from raytracing_amd import api, host
    api.init(0)
    g = _load(golden_dir, sid)
    dev = host.HostScene.builtin(sid).upload()   # scene built by the product's C++ host code
    hits = dev.intersect(g["org"], g["dir"])
    assert hits.tobytes() == g["hits"].tobytes()
    rgb, st = dev.trace(g["t_org"], g["t_dir"], g["t_pixel"], g["t_sample"], seed=1, k0=2)
    assert [st.rays, st.samples, st.nodes] == list(g["t_counts"])
    assert np.abs(rgb - g["t_rgb"]).max() <= 5e-6          # colours differ only through powf (double power rounded once vs glibc)
    img, ist = dev.render(host.camera_default(48, 27), 48, 27, 2, seed=3, tile_w=16, tile_h=8)
    assert [ist.rays, ist.samples, ist.nodes] == list(g["img_counts"])
    assert np.abs(img - g["img"]).max() <= 5e-6
    dev.close()
