#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Usage: python tools/pmc_traffic.py fetch_results.db write_results.db batch > profiles/rNN_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB. On gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads, so fetched bytes = 2 * FETCH_SIZE * 1024 (exact for 16 B/lane reads,
an upper estimate for narrower ones); WRITE_SIZE is taken as is.
"""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
    ci = {c: i for i, c in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ci else "name"
    tot, disp = defaultdict(float), defaultdict(set)
    for r in cur.execute("select * from counters_collection"):
        k = str(r[ci[name_col]]).split("(")[0].replace("void ", "").split("<")[0]
        if not k.startswith("k_") or r[ci["counter_name"]] != counter:
            continue
        tot[k] += float(r[ci["value"]])
        disp[k].add(r[ci["dispatch_id"]])
    return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"frames_per_launch": int(sys.argv[3]), "note": "bytes = 2*FETCH_SIZE*1024 (gfx950 half-count correction) + WRITE_SIZE*1024, averaged over the launches of a kernel",
       "kernels": {}}
for k in sorted(fetch):
    f_kb, n = fetch[k]
    w_kb = write.get(k, (0.0, 0))[0]
    out["kernels"][k] = {"launches_profiled": n, "fetch_size_kb_per_launch": round(f_kb, 1), "write_size_kb_per_launch": round(w_kb, 1),
                         "hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024)}
print(json.dumps(out, indent=1))
