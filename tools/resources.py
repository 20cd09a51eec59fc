#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel of raytracing_amd/csrc/rt_api.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (`make -C raytracing_amd/csrc resources`; cross-compiles, no GPU needed).
usage: resources.py [--filter k_render] > profiles/rN_resources.txt"""
import argparse
import os
import re
import subprocess

ap = argparse.ArgumentParser()
ap.add_argument("--filter", default="")
ap.add_argument("--extra", default="", help="EXTRA flags for the Makefile, e.g. -DRT_SB0=4")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["make", "-C", os.path.join(root, "raytracing_amd", "csrc"), "resources"] + ([f"EXTRA={a.extra}"] if a.extra else [])
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line) or re.search(r"Name: (\S+) \[", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950" + (f", EXTRA={a.extra}" if a.extra else ""))
print(f"{'kernel':62s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>14s} {'VGPR spills':>11s} {'SGPR spills':>11s} {'waves/SIMD':>10s} {'LDS B':>7s}")
for r in rows:
    if a.filter and a.filter not in r["name"]:
        continue
    name = re.sub(r"\(.*", "", r["name"]).replace("void ", "")
    print(f"{name[:62]:62s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} {r.get('ScratchSize', 0):14d} "
          f"{r.get('VGPRs Spill', 0):11d} {r.get('SGPRs Spill', 0):11d} {r.get('Occupancy', 0):10d} {r.get('LDS Size', 0):7d}")
