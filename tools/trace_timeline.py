#!/usr/bin/env python3
"""Timeline of the extraction kernels of the last pass in a rocprofv3 kernel trace (rocpd sqlite database):
start and end of every dispatch relative to the pass's first kernel, its stream / queue, and the idle gaps.

Usage: python tools/trace_timeline.py name_results.db [passes_from_end]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = "name, start, end" + (", " + qcol if qcol else "")
rows = [r for r in cur.execute("select %s from kernels order by start" % sel) if "k_" in r[0]]
# a pass ends with k_describe
ends = [i for i, r in enumerate(rows) if "k_describe" in r[0]]
if len(ends) < back + 1:
    sys.exit("not enough passes in the trace")
lo, hi = ends[-back - 1] + 1, ends[-back]
t0 = rows[lo][1]
prev_end = t0
print("# kernel, queue, start us, end us, duration us (relative to the first kernel of the pass)")
for r in rows[lo:hi + 1]:
    name = r[0].split("(")[0].replace("void ", "")
    print("%-28s q=%-4s %9.1f %9.1f %9.1f" % (name[:28], r[3] if qcol else "-", (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3))
print("# pass: %.1f us from the first start to the last end; next pass starts %.1f us after this one's end"
      % ((rows[hi][2] - t0) / 1e3, ((rows[hi + 1][1] - rows[hi][2]) / 1e3) if hi + 1 < len(rows) else float("nan")))
