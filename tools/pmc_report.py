#!/usr/bin/env python3
"""Per-kernel table from the counter passes of tools/pmc_round.sh.

Usage: python tools/pmc_report.py gpurun_out/<tag> <prefix>      (prefix: ext, st or m)

Reads <prefix>_{lds,issue,clk,fetch,write}.txt (per-kernel counter sums, tools/pmc_summary.py) and
<prefix>_trace.txt (kernel durations, tools/prof_summary.py) and prints, per kernel and launch:
  time            average duration of a launch (kernel trace)
  clock           GRBM_GUI_ACTIVE / 8 XCDs / duration
  valu_busy       SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (1024 SIMDs * cycles)
  lds_busy        SQ_LDS_IDX_ACTIVE / (256 CUs * cycles)
  lds_conflict    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE  (share of the LDS-array cycles that are conflict cycles)
  hbm_bytes       2 * FETCH_SIZE KB (gfx950 counts half of wide streaming reads: exact for 16 B/lane, an upper
                  estimate for narrower accesses) + WRITE_SIZE KB, and the rate hbm_bytes / time
"""
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

d, prefix = Path(sys.argv[1]), sys.argv[2]


def counters(name):
    out = defaultdict(dict)
    f = d / ("%s_%s.txt" % (prefix, name))
    if not f.exists():
        return out
    k = None
    for line in f.read_text().splitlines():
        m = re.match(r"^(\S.*?) dispatches=(\d+)", line)
        if m:
            k = m.group(1).split("<")[0]
            continue
        m = re.match(r"^\s+(\S+)\s+\d+\s+per dispatch\s+([\d.]+)", line)
        if m and k:
            out[k][m.group(1)] = out[k].get(m.group(1), 0.0) + float(m.group(2))   # template variants of one kernel add up
    return out


def durations():
    out = {}
    f = d / ("%s_trace.txt" % prefix)
    if not f.exists():
        return out
    for line in f.read_text().splitlines():
        m = re.match(r"^(?:void )?(k_\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            k = m.group(1).split("<")[0]
            calls, total = int(m.group(2)), float(m.group(3))
            c0, t0 = out.get(k, (0, 0.0))
            out[k] = (c0 + calls, t0 + total)
    return {k: t / c for k, (c, t) in out.items()}          # average microseconds per launch


lds, issue, clk, fetch, write, dur = (counters("lds"), counters("issue"), counters("clk"), counters("fetch"), counters("write"), durations())
report = {}
for k in sorted(set(lds) | set(issue)):
    r = {}
    us = dur.get(k)
    cyc = clk.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if us:
        r["avg_launch_us"] = round(us, 1)
        if cyc:
            r["clock_GHz"] = round(cyc / us / 1e3, 2)
    if cyc:
        r["valu_busy"] = round(issue.get(k, {}).get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (1024 * cyc), 3)
        r["lds_busy"] = round(lds.get(k, {}).get("SQ_LDS_IDX_ACTIVE", 0.0) / (256 * cyc), 3)
    act = lds.get(k, {}).get("SQ_LDS_IDX_ACTIVE", 0.0)
    r["lds_conflict_share"] = round(lds.get(k, {}).get("SQ_LDS_BANK_CONFLICT", 0.0) / act, 3) if act else 0.0
    r["valu_insts"] = int(issue.get(k, {}).get("SQ_INSTS_VALU", 0))
    r["salu_insts"] = int(issue.get(k, {}).get("SQ_INSTS_SALU", 0))
    r["lds_insts"] = int(lds.get(k, {}).get("SQ_INSTS_LDS", 0))
    fb = 2 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
    wb = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    r["hbm_fetch_bytes"], r["hbm_write_bytes"] = int(fb), int(wb)
    if us:
        r["hbm_GBps"] = round((fb + wb) / us / 1e3, 1)
    report[k] = r
print(json.dumps(report, indent=1))
