#!/usr/bin/env python3
"""Vector-ALU instructions per kernel launch from a rocprofv3 --pmc pass that holds SQ_INSTS_VALU and
SQ_WAVES (e.g. the SQ pass of tools/profile_round.sh).

Usage: python tools/pmc_valu.py sq_results.db frames_per_launch > profiles/rNN_valu.json

A wave64 vector instruction occupies its SIMD for 4 cycles, so a chip with 1024 SIMDs issues at most
1024 / 4 of them per cycle; bench.py turns the per-launch count into a fraction of that rate.
"""
import json
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
ci = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ci else "name"
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in cur.execute("select * from counters_collection"):
    k = str(r[ci[name_col]]).split("(")[0].replace("void ", "").split("<")[0]
    if not k.startswith("k_"):
        continue
    tot[k][r[ci["counter_name"]]] += float(r[ci["value"]])
    disp[k].add(r[ci["dispatch_id"]])
out = {"frames_per_launch": int(sys.argv[2]), "kernels": {}}
for k in sorted(tot):
    n = len(disp[k])
    out["kernels"][k] = {"launches_profiled": n, "valu_insts_per_launch": int(tot[k].get("SQ_INSTS_VALU", 0) / n),
                         "waves_per_launch": int(tot[k].get("SQ_WAVES", 0) / n)}
print(json.dumps(out, indent=1))
