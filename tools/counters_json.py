#!/usr/bin/env python3
"""profiles/counters_<config>.json from a directory of rocprofv3 --pmc passes of ONE frame (tools/prof_round3.sh): raw counter
sums of the render kernels plus the figures derived from them, stamped with the hash of the kernel sources so that bench.py
only reports them for the build they were measured on.
  fetch_bytes / write_bytes   FETCH_SIZE / WRITE_SIZE x 1024 (the counters report KB): memory-side requests of the L2,
                              Infinity-Cache hits included; gfx950 tallies a 128-B read request as 64 B (not corrected here)
  l2_hit_rate                 TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); TCC_REQ_sum counts 128-B requests
  valu_issue_share            SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): a SIMD issues one wave64 VALU
                              instruction per 2 cycles at best; GRBM_GUI_ACTIVE is summed over the 8 XCDs
  active_lanes                SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, of 64
  wait_inst_any_share, wait_any_share   SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES
usage: counters_json.py <pmc dir> <config> [--traced-rays N] [--frame-ms T] [--out profiles/counters_cX.json]"""
import argparse
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("config")
ap.add_argument("--traced-rays", type=float, default=0)
ap.add_argument("--frame-ms", type=float, default=0)
ap.add_argument("--out", default="")
a = ap.parse_args()
c = {}
for db in sorted(glob.glob(a.dir + "/*_results.db")):
    cur = sqlite3.connect(db).cursor()
    try:
        for name, val in cur.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%k_render%' group by counter_name"):
            c[name] = val
    except sqlite3.OperationalError:
        pass
rec = {"config": a.config, "kernel": "k_render (all launches of one frame)", "kernel_source_hash": bench.kernel_source_hash(),
       "fetch_bytes": c["FETCH_SIZE"] * 1024.0, "write_bytes": c["WRITE_SIZE"] * 1024.0,
       "how": "one rocprofv3 --pmc run per counter set over one frame (tools/prof_round3.sh)"}
for k in sorted(c):
    if k not in ("FETCH_SIZE", "WRITE_SIZE"):
        rec[k] = c[k]
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    rec["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
    rec["valu_issue_share"] = round(c["SQ_INSTS_VALU"] * 2 / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 4)
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    rec["active_lanes"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 2)
if "SQ_WAVE_CYCLES" in c:
    for k, name in (("SQ_WAIT_INST_ANY", "wait_inst_any_share"), ("SQ_WAIT_ANY", "wait_any_share")):
        if k in c:
            rec[name] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
if a.traced_rays and "SQ_INSTS_VALU" in c:
    rec["valu_insts_per_64_traced_rays"] = round(c["SQ_INSTS_VALU"] * 64 / a.traced_rays, 1)
if a.frame_ms:
    rec["profiled_frame_ms"] = a.frame_ms
out = a.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"counters_{a.config}.json")
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec))
