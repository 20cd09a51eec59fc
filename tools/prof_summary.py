#!/usr/bin/env python3
"""Turn a rocprofv3 results database (rocpd sqlite, ROCm 7.x default output) into the text summary
kept under profiles/: per-kernel calls / total / mean duration, and per-dispatch resources of the
kernels whose name matches --kernel.   usage: prof_summary.py results.db [--kernel k_render]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--kernel", default="k_")
ap.add_argument("--cmd", default="")
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
if a.cmd:
    print(f"# command: {a.cmd}")
print("# kernel stats (rocprofv3 --kernel-trace --stats): name | calls | total_ms | mean_ms | percent")
for name, calls, total, mean, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name[:110]:110s} | {calls:6d} | {total / 1e3:14.3f} | {mean / 1e3:12.4f} | {pct:7.3f}")
print(f"\n# dispatches of kernels matching '{a.kernel}': name | duration_ms | arch_vgpr | accum_vgpr | sgpr | scratch_B/lane | lds | grid | workgroup")
q = ("select name,duration,vgpr_count,accum_vgpr_count,sgpr_count,scratch_size,lds_size,grid_x,workgroup_x from kernels "
     "where name like ? order by start")
for r in cur.execute(q, (f"%{a.kernel}%",)):
    print(f"{r[0][:60]:60s} | {r[1] / 1e6:12.4f} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} | {r[8]}")
