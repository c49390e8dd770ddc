#!/usr/bin/env python3
"""Per-dispatch view of a PMC pass for one kernel: python tools/pmc_rounds.py <dir> <kernel substring>"""
import csv, glob, os, sys, collections
d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    e = disp.setdefault(r["Dispatch_Id"], {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "grid": r.get("Grid_Size")})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in list(disp.items())[-int(sys.argv[3]) if len(sys.argv) > 3 else 0:]:
    print(k, {a: (round(b, 3) if isinstance(b, float) and b < 1e4 else (f"{b:.4g}" if isinstance(b, float) else b)) for a, b in v.items()})
