"""print the headline and side entries of a bench.py JSON line (tools/gpu_job6.sh)"""
import json
import sys

for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        r = d["roofline"]
        print("line", d["metric"][:40], "value", round(d["value"]), "frac", round(r["frac"], 4), "sustained", r.get("sustained_frac"),
              "us_per_layer", r.get("us_per_layer"), "us_per_step", r.get("us_per_step"))
        if d.get("side"):
            print("side", json.dumps(d["side"])[:2500])
