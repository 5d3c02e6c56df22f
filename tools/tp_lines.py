#!/usr/bin/env python3
"""print the timing fields of bench.py tensor-parallel lines (plumbing runs)"""
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith('{"metric"'):
            d = json.loads(l); r = d["roofline"]
            print(f, "| us_per_step", round(r["us_per_step"], 1), "| gemv us_per_launch", round(r["us_per_launch"], 1), "| collective_us_per_step",
                  round(r["collective_us_per_step"], 1), "|", d["config"]["parallelism"][:140])
