#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 --pmc results database.

Usage: python tools/pmc_summary.py gpurun_out/dir/name_results.db [kernel-prefix]
"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
prefix = sys.argv[2] if len(sys.argv) > 2 else "k_"
cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
rows = cur.execute("select * from counters_collection").fetchall()
ci = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ci else "name"
agg = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(set)
for r in rows:
    k = str(r[ci[name_col]]).split("(")[0].replace("void ", "")
    if not k.startswith(prefix):
        continue
    agg[k][r[ci["counter_name"]]] += float(r[ci["value"]])
    ndisp[k].add(r[ci["dispatch_id"]])
for k in agg:
    print(k, "dispatches=%d" % len(ndisp[k]))
    for c, v in sorted(agg[k].items()):
        print("   %-28s %16.0f   per dispatch %14.1f" % (c, v, v / len(ndisp[k])))
