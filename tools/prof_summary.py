#!/usr/bin/env python3
"""Turns a rocprofv3 results database (rocpd sqlite, `--kernel-trace --stats`) into a text summary.

Usage: python tools/prof_summary.py gpurun_out/prof_x/name_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
print("# per kernel: calls, total us, average us, percent of kernel time")
print("%-14s %7s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in cur.execute(
        "select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-14s %7d %14.1f %12.2f %8.2f" % (name.split("(")[0][-40:], calls, total / 1000.0 if total > 1e6 else total, avg / 1000.0 if total > 1e6 else avg, pct))
print()
print("# resources per kernel (first dispatch): vgpr, accum_vgpr, sgpr, lds bytes, scratch bytes, workgroup")
seen = set()
for row in cur.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, workgroup_y "
                       "from kernels order by start"):
    k = row[0].split("(")[0]
    if k in seen:
        continue
    seen.add(k)
    print("%-40s vgpr=%d agpr=%d sgpr=%d lds=%d scratch=%d wg=%dx%d" % ((k[-40:],) + tuple(row[1:])))
if "-v" in sys.argv:
    print()
    print("# dispatches of k_* kernels: name grid duration_ns")
    for row in cur.execute("select name, grid_x, grid_y, grid_z, duration from kernels where name like 'k_%' order by start"):
        print("%-12s grid=%dx%dx%d %d" % (row[0].split("(")[0], row[1], row[2], row[3], row[4]))
