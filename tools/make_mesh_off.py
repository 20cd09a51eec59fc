#!/usr/bin/env python3
"""Writes a synthetic triangle mesh as an OFF file (the format Mesh::loadOFF reads): a bumpy n x n vertex
height field folded into a dome, 2 (n-1)^2 triangles.  n = 225 gives 100 352 triangles, n = 709 about 1 M.
usage: make_mesh_off.py out.off [n] [seed]"""
import sys

import numpy as np


def make(n, seed=1):
    rng = np.random.default_rng(seed)
    u = np.linspace(-1.0, 1.0, n)
    gx, gz = np.meshgrid(u, u, indexing="xy")
    r2 = gx * gx + gz * gz
    gy = 0.6 * np.exp(-1.5 * r2) + 0.04 * rng.random((n, n)) + 0.05 * np.sin(9 * gx) * np.cos(7 * gz)
    pos = np.stack([gx, gy, gz], -1).reshape(-1, 3)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    tri = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)])
    return pos, tri


def write_off(path, pos, tri):
    with open(path, "w") as f:
        f.write(f"OFF\n{len(pos)} {len(tri)} 0\n\n")
        np.savetxt(f, pos, fmt="%.7g")
        np.savetxt(f, np.concatenate([np.full((len(tri), 1), 3), tri], 1), fmt="%d")


if __name__ == "__main__":
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 225
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    p, t = make(n, seed)
    write_off(out, p, t)
    print(f"{out}: {len(p)} vertices, {len(t)} triangles")
