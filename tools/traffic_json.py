#!/usr/bin/env python3
"""profiles/traffic_<config>.json from a directory of rocprofv3 --pmc passes (tools/pmc_config.sh, tools/pmc_traffic.sh):
FETCH_SIZE / WRITE_SIZE of the render kernels summed over the launches of one frame, in bytes (the counters report KB),
stamped with the hash of the kernel sources so that bench.py only reports the figure for the build it was measured on.
usage: traffic_json.py <pmc dir> <config> [--out profiles/traffic_cX.json]"""
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
rec = {"config": a.config, "kernel": "k_render (all launches of one frame)", "fetch_bytes": c["FETCH_SIZE"] * 1024.0, "write_bytes": c["WRITE_SIZE"] * 1024.0,
       "kernel_source_hash": bench.kernel_source_hash(),
       "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of one frame; counters report KB of memory-side (fabric) requests of the L2, "
              "Infinity-Cache hits included; on gfx950 FETCH_SIZE tallies a 128-B request as 64 B for wide coalesced reads (not corrected here)"}
for k in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
    if k in c:
        rec[k] = c[k]
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    rec["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
out = a.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"traffic_{a.config}.json")
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec))
