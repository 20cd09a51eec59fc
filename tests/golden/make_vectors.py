"""Writes tests/golden/vectors_scene{0..3}.npz: seeded inputs and the outputs of the pinned CPU oracle
(oracle/rt_oracle.cpp, which reproduces the reference's rendu.ppm byte for byte) for the four built-in
scenes.  The reference itself cannot be run in the build container (no GLUT headers), so these vectors are
oracle outputs, not reference outputs; they freeze the oracle against accidental change and give the GPU
tests a committed target.

  org, dir      1 584 rays (stored Ray fields: unit directions): jittered primary rays, interior, outside,
                axis-parallel and sphere-surface rays
  hits          rt_hit records of Scene::computeIntersection for them
  t_org/t_dir/t_pixel/t_sample, t_rgb   320 primary rays traced with Scene::rayTrace under the counter RNG (seed 1, k0 2)
  img           48 x 27 x 2 spp render (seed 3) with ray / node counts
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orc  # noqa: E402
from tests import rayset  # noqa: E402


def inputs():
    org, d = rayset.mixed(orc, n_each=200)
    keep = np.r_[0:584:1, 5184:5184 + 1000]  # thin the 96x54 primary grid, keep the 5 x 200 special rays
    org, d = org[keep], d[keep]
    w, h, spp = 16, 10, 2
    rng = np.random.default_rng(77)
    ys, xs = np.mgrid[0:h, 0:w]
    pix = np.repeat((xs + ys * w).ravel().astype(np.uint32), spp)
    smp = np.tile(np.arange(spp, dtype=np.uint32), w * h)
    u = ((np.repeat(xs.ravel(), spp) + rng.random(pix.size)) / w).astype(np.float32)
    v = ((np.repeat(ys.ravel(), spp) + rng.random(pix.size)) / h).astype(np.float32)
    t_org, t_dir = orc.generate_rays(orc.camera_default(w, h), u, v)
    return org, d, t_org, orc.normalize_dirs(t_dir), pix, smp


def outputs(scene_id, org, d, t_org, t_dir, pix, smp):
    s = orc.Scene.builtin(scene_id)
    hits = s.intersect(org, d)
    rgb, st = s.trace(t_org, t_dir, pix, smp, seed=1, k0=2)
    img, ist = s.render(orc.camera_default(48, 27), 48, 27, 2, orc.RT_RNG_COUNTER, seed=3)
    return dict(hits=hits, t_rgb=rgb, t_counts=np.array([st.rays, st.samples, st.nodes], np.uint64), img=img,
                img_counts=np.array([ist.rays, ist.samples, ist.nodes], np.uint64))


if __name__ == "__main__":
    org, d, t_org, t_dir, pix, smp = inputs()
    for sid in range(4):
        out = outputs(sid, org, d, t_org, t_dir, pix, smp)
        np.savez_compressed(os.path.join(HERE, f"vectors_scene{sid}.npz"), org=org, dir=d, t_org=t_org, t_dir=t_dir, t_pixel=pix,
                            t_sample=smp, **out)
        print(sid, int(out["hits"]["hit"].sum()), "hits;", out["t_counts"], out["img_counts"])
