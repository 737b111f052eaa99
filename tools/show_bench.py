#!/usr/bin/env python3
"""Prints a bench.py JSON line key by key (long values cut). Usage: python tools/show_bench.py file.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms per step")
for k, v in d["roofline"].items():
    print("  roofline.%s: %s" % (k, json.dumps(v)[:600]))
for k in d:
    if k not in ("roofline", "config", "value", "unit", "ms_per_step"):
        print("  %s: %s" % (k, json.dumps(d[k])[:400]))
