#!/usr/bin/env python3
"""Writes the synthetic dome mesh of raytracing_amd.workloads as an OFF file (the format Mesh::loadOFF reads).
n = 225 gives 100 352 triangles, n = 709 about 1 M.   usage: make_mesh_off.py out.off [n] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.workloads import dome_mesh as make, write_off  # noqa: E402,F401

if __name__ == "__main__":
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 225
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    p, t = make(n, seed)
    write_off(out, p, t)
    print(f"{out}: {len(p)} vertices, {len(t)} triangles")
