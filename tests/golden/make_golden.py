"""Regenerates tests/golden/ from data files the reference ships (run in the build container,
where /root/reference exists; the GPU box only sees the committed outputs).

  rendu_scene3_480x480_50spp.npz  decoded RGB bytes of R/rendu.ppm (ASCII P3), the render the reference
                                  leaves next to its sources
  rendu_scene2_480x480_50spp.npz  decoded RGB bytes of R/Compte-rendu/Images/Phase4/renduSphereVerre2.png, the
                                  report's final render of scene 2 (Cornell box, mirror + glass sphere): the
                                  same 480x480 / 50 spp / never-seeded rand() flow, saved as PNG (needs PIL)
  star.off                        R/img/star.off, the mesh scene 3 loads (input data, 14 v / 24 t)
"""
import os
import shutil

import numpy as np

R = "/root/reference/HAI719I_Raytracer"
HERE = os.path.dirname(os.path.abspath(__file__))

toks = open(os.path.join(R, "rendu.ppm")).read().split()
assert toks[0] == "P3"
w, h, mx = (int(t) for t in toks[1:4])
rgb = np.array(toks[4:], dtype=np.uint8).reshape(h, w, 3)
np.savez_compressed(os.path.join(HERE, "rendu_scene3_480x480_50spp.npz"), rgb=rgb)
shutil.copyfile(os.path.join(R, "img", "star.off"), os.path.join(HERE, "star.off"))

from PIL import Image  # noqa: E402

png = np.array(Image.open(os.path.join(R, "Compte-rendu", "Images", "Phase4", "renduSphereVerre2.png")))
assert png.shape == (480, 480, 4) and (png[..., 3] == 255).all()
np.savez_compressed(os.path.join(HERE, "rendu_scene2_480x480_50spp.npz"), rgb=np.ascontiguousarray(png[..., :3]))
