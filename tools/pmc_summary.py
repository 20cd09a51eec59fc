#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (rocpd sqlite) of the render kernel: raw counter sums plus the
derived figures (instructions per ray, issue utilisation, waits).  usage: pmc_summary.py <dir with p*_results.db> --rays N --ms T"""
import argparse
import glob
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--kernel", default="k_render")
ap.add_argument("--rays", type=float, default=0)
ap.add_argument("--ms", type=float, default=0)
a = ap.parse_args()
c = {}
for db in sorted(glob.glob(a.dir + "/*_results.db")):
    cur = sqlite3.connect(db).cursor()
    for name, val in cur.execute("select counter_name, sum(value) from counters_collection where kernel_name like ? group by counter_name", (f"%{a.kernel}%",)):
        c[name] = val
for k in sorted(c):
    print(f"{k:28s} {c[k]:.6e}")
if a.rays and "SQ_INSTS_VALU" in c:
    print(f"\nVALU wave-instructions per 64 rays : {c['SQ_INSTS_VALU'] * 64 / a.rays:.1f}")
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        print(f"mean active lanes per VALU instr   : {c['SQ_THREAD_CYCLES_VALU'] / c['SQ_ACTIVE_INST_VALU']:.1f} of 64 (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU)")
if a.ms and "GRBM_GUI_ACTIVE" in c:
    clk = c["GRBM_GUI_ACTIVE"] / 8 / (a.ms * 1e-3)
    print(f"shader clock                       : {clk / 1e9:.2f} GHz")
    simd_cycles = clk * a.ms * 1e-3 * 1024
    if "SQ_INSTS_VALU" in c:
        # a SIMD issues one wave64 VALU instruction per 2 cycles at best (measured with tools/ubench/valu_rate.hip)
        print(f"VALU issue-slot utilisation (SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * cycles)) : {c['SQ_INSTS_VALU'] * 2 / simd_cycles * 100:.1f} %")
if "SQ_WAVE_CYCLES" in c:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in c:
            print(f"{k} / SQ_WAVE_CYCLES = {c[k] / c['SQ_WAVE_CYCLES'] * 100:.1f} %")
