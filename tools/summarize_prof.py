#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs written by tools/profile.sh: per-kernel duration stats
from the kernel trace and per-dispatch averages of every PMC counter."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def kernel_stats(d):
    rows = []
    for f in find(d, "*kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    agg = defaultdict(list)
    for r in rows:
        try:
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        except (KeyError, ValueError):
            continue
        agg[r.get("Kernel_Name", "?")].append(dur)
    out = []
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        out.append(dict(kernel=k[:100], calls=len(v), total_us=sum(v) / 1e3, avg_us=sum(v) / len(v) / 1e3,
                        med_us=v2[len(v2) // 2] / 1e3, min_us=v2[0] / 1e3, max_us=v2[-1] / 1e3))
    return out


def pmc_stats(d, want="pbl_gemv"):
    agg = defaultdict(lambda: defaultdict(list))
    for f in find(d, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "?")
                if want not in k:
                    continue
                try:
                    agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                except (KeyError, ValueError):
                    pass
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    root = sys.argv[1]
    print(f"# rocprofv3 summary for {root}")
    tr = os.path.join(root, "trace")
    if os.path.isdir(tr):
        print("\n## kernel trace (durations in us)")
        for s in kernel_stats(tr)[:8]:
            print(json.dumps(s))
        for f in find(tr, "*kernel_stats.csv")[:1]:
            print(f"\n## {os.path.basename(f)} (rocprofv3 --stats)")
            print("".join(open(f).readlines()[:8]))
    for sub in sorted(os.listdir(root)):
        p = os.path.join(root, sub)
        if sub.startswith("pmc") and os.path.isdir(p):
            st = pmc_stats(p)
            if st:
                print(f"\n## {sub}: per-dispatch averages")
                for k, cs in st.items():
                    print(k, json.dumps({c: round(v, 1) for c, v in sorted(cs.items())}))
            ks = [s for s in kernel_stats(p) if "pbl_gemv" in s["kernel"]]
            for s in ks[:2]:
                print("   (profiled-run duration)", json.dumps(s))
    for log in sorted(glob.glob(os.path.join(root, "*.log"))):
        for line in open(log):
            if line.startswith("{\"metric\""):
                j = json.loads(line)
                print(f"\n## bench line under {os.path.basename(log)}: value={j['value']:.0f} "
                      f"achieved={j['roofline']['achieved']:.0f} GB/s us/launch={j['roofline']['us_per_launch']:.1f}")


if __name__ == "__main__":
    main()
