#!/usr/bin/env python3
"""Per-kernel table from the counter passes of tools/pmc_round.sh.

Usage: python tools/pmc_report.py gpurun_out/<tag> <prefix> [frames_per_pass out_prefix]     (prefix: ext, st or m)

With the two optional arguments the per-pass HBM bytes and vector-instruction counts are also written as
<out_prefix>_traffic.json and <out_prefix>_valu.json, the summaries bench.py reads for its roofline object.

Reads <prefix>_{lds,issue,clk,fetch,write}.txt (per-kernel counter sums, tools/pmc_summary.py) and
<prefix>_trace.txt (kernel durations, tools/prof_summary.py) and prints, per kernel and PASS (one pass =
one batch of frames through the pipeline = one launch of k_describe / k_stereo_match / the matcher; the
template variants of a kernel and its launches per pass -- seven k_resize, two k_fast -- are added up):
  time            duration per pass (kernel trace; <prefix>_trace_alone.txt, kernels one at a time, where it exists)
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


def family(kernel):
    return kernel.replace("void ", "").split("<")[0]


def counters(name):
    """{family: {counter: total over all dispatches}}, {family: dispatches}"""
    out, ndisp = defaultdict(dict), defaultdict(int)
    f = d / ("%s_%s.txt" % (prefix, name))
    if not f.exists():
        return out, ndisp
    k = None
    for line in f.read_text().splitlines():
        m = re.match(r"^(\S.*?) dispatches=(\d+)", line)
        if m:
            k = family(m.group(1))
            ndisp[k] += int(m.group(2))
            continue
        m = re.match(r"^\s+(\S+)\s+(\d+)\s+per dispatch\s+([\d.]+)", line)
        if m and k:
            out[k][m.group(1)] = out[k].get(m.group(1), 0.0) + float(m.group(2))
    return out, ndisp


def per_pass(pair):
    """Counter totals divided by the number of passes of that run (the launches of the once-per-pass kernel)."""
    tot, ndisp = pair
    once = [ndisp[k] for k in ("k_describe", "k_stereo_median", "k_search_proj", "k_search_map") if ndisp.get(k)]
    passes = min(once) if once else 1
    return defaultdict(dict, {k: {c: v / passes for c, v in cs.items()} for k, cs in tot.items()})


def durations():
    calls, total = defaultdict(int), defaultdict(float)
    f = d / ("%s_trace_alone.txt" % prefix)                   # one kernel at a time, as the counter passes run them
    if not f.exists():
        f = d / ("%s_trace.txt" % prefix)
    if not f.exists():
        return {}
    for line in f.read_text().splitlines():
        m = re.match(r"^(?:void )?(k_\S+?(?:<[^>]*>)?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            k = family(m.group(1))
            calls[k] += int(m.group(2))
            total[k] += float(m.group(3))
    once = [calls[k] for k in ("k_describe", "k_stereo_median", "k_search_proj", "k_search_map") if calls.get(k)]
    passes = min(once) if once else 1
    return {k: total[k] / passes for k in total}                # microseconds per pass


lds, issue, clk, fetch, write, dur = (per_pass(counters("lds")), per_pass(counters("issue")), per_pass(counters("clk")), per_pass(counters("fetch")),
                                      per_pass(counters("write")), durations())
report = {}
for k in sorted(set(lds) | set(issue)):
    r = {}
    us = dur.get(k)
    cyc = clk.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if us:
        r["us_per_pass"] = round(us, 1)
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
if len(sys.argv) >= 5:
    frames, outp = int(sys.argv[3]), sys.argv[4]
    src = "%s/%s_{fetch,write,issue}.txt" % (d, prefix)
    traffic = {"frames_per_launch": frames,
               "note": "per pass (all launches of a kernel for one batch of frames): bytes = 2*FETCH_SIZE*1024 (gfx950 half-count "
                       "correction: exact for 16 B/lane reads, an upper estimate for narrower ones) + WRITE_SIZE*1024 (" + src + ")",
               "kernels": {k: {"hbm_bytes_per_launch": r["hbm_fetch_bytes"] + r["hbm_write_bytes"], "fetch_bytes": r["hbm_fetch_bytes"],
                               "write_bytes": r["hbm_write_bytes"]} for k, r in report.items()}}
    valu = {"frames_per_launch": frames, "note": "SQ_INSTS_VALU per pass (" + src + ")",
            "kernels": {k: {"valu_insts_per_launch": r["valu_insts"]} for k, r in report.items()}}
    Path(outp + "_traffic.json").write_text(json.dumps(traffic, indent=1))
    Path(outp + "_valu.json").write_text(json.dumps(valu, indent=1))
