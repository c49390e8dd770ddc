#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass (tools/pmc_pass.sh): prints and writes <dir>/digest.json."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dvp::", "")
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r.get("End_Timestamp"):
        dur[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
out = {}
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    tot = sum(dur[k].values())
    out[k] = {"launches": n, "total_ms": tot, "avg_ms": tot / max(len(dur[k]), 1)}
    for c, v in cs.items():
        out[k][c] = sum(v) / len(v)
top = sorted(out.items(), key=lambda kv: -kv[1]["total_ms"])[:int(os.environ.get("PMC_TOP", "12"))]
for k, v in top:
    print(k[:40], json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()}))
json.dump(out, open(os.path.join(d, "digest.json"), "w"), indent=1)
