"""Tries to pin the oracle's point-light branch (scenes 0 and 1: setup_single_sphere / setup_single_square,
R/src/Scene.h:722-807, phong's LightType_Spherical branch :396-415) with the screenshots of the reference's report:
renders both scenes with the oracle the way the reference does (480 x 480, 50 spp, never-seeded glibc rand()) and counts,
for every 480 x 480 PNG under R/Compte-rendu/Images/Phase1..3, the pixels that agree byte for byte.  Run in the build
container (needs /root/reference and PIL); prints one line per image.  An image that matches everywhere would become a
known-answer test; the output of the last run is recorded in DESIGN.md."""
import glob
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import orc  # noqa: E402

R = "/root/reference/HAI719I_Raytracer"
renders = {}
for sid in (0, 1):
    img, _ = orc.Scene.builtin(sid).render(orc.camera_default(480, 480), 480, 480, 50, orc.RT_RNG_LIBC, 1)
    renders[sid] = orc.quantize(img)
best = (0.0, None, None)
for path in sorted(glob.glob(os.path.join(R, "Compte-rendu", "Images", "Phase[123]", "*.png"))):
    png = np.array(Image.open(path).convert("RGB"))
    if png.shape != (480, 480, 3):
        print(f"{os.path.relpath(path, R)}: {png.shape[1]}x{png.shape[0]}, not a 480x480 render")
        continue
    line = []
    for sid, q in renders.items():
        same = float((png == q).all(axis=2).mean())
        close = float((np.abs(png.astype(int) - q.astype(int)).max(axis=2) <= 2).mean())
        line.append(f"scene {sid}: {same * 100:6.2f} % equal, {close * 100:6.2f} % within 2 levels")
        if same > best[0]:
            best = (same, path, sid)
    print(f"{os.path.relpath(path, R)}: " + "; ".join(line))
print("best:", best)
