#!/usr/bin/env python3
"""Turns gpurun_out/refresh/ (tools/refresh_profiles.sh) into the committed summaries under profiles/.
usage: python tools/digest_profiles.py rNN"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
KERNEL = "k_affine_round<true"


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    assert hits, pattern
    return hits[0]


def last_json_line(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, f"{tag}_bench_prove2p20.json"))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, f"{tag}_bench_prove2p20_under_rocprof.json"))
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_bench_prove2p20_kernel_stats.csv"))
shutil.copy(one("msm/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_msm_kernel_stats.csv"))


def pmc(dirname, counter):
    rows = list(csv.DictReader(open(one(f"{dirname}/**/*counter_collection.csv"))))
    vals, durs = [], []
    for r in rows:
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return vals, durs


fetch, d1 = pmc("pmc_fetch", "FETCH_SIZE")
write, d2 = pmc("pmc_write", "WRITE_SIZE")
assert fetch and write and len(fetch) == len(write), (len(fetch), len(write))
avg_f, avg_w = sum(fetch) / len(fetch), sum(write) / len(write)
out = {
    "kernel": "dvp::k_affine_round<true, B> (B = 32 additions per inversion in these launches, 16 in small rounds)",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
    "launches": len(fetch),
    "avg_FETCH_SIZE_KB": avg_f,
    "avg_WRITE_SIZE_KB": avg_w,
    "avg_duration_ms_under_pmc": (sum(d1) / len(d1)) if d1 else None,
    "calibration": "expected per launch (average of the 2m- and 4m-point MSMs, ~20M pair additions): two 64-B bases gathered in pass 1 "
                   "and again in pass 2, 8 B of indices, 32 B prefix written + read, 64 B output; the counters are taken as-is (no x2: "
                   "MI355X_MICROARCH.md's half-counting was measured on wide coalesced streams, these are 64-B gathers, and the un-doubled "
                   "value already matches the expected byte count); Infinity-Cache hits of the pass-2 re-reads are included by the counter",
    "traffic_bytes_per_launch": (avg_f + avg_w) * 1024.0,
}
json.dump(out, open(os.path.join(DST, f"{tag}_pmc_traffic_k_affine_round0.json"), "w"), indent=1)
b = last_json_line(os.path.join(SRC, "bench.json"))
print("bench:", b["value"], "constraints/s", b["ms_per_step"], "ms; roofline", b["roofline"]["avg_launch_ms"], "ms/launch, work_model frac",
      b["roofline"]["work_model"]["frac"])
print("traffic per launch: %.3f GB" % (out["traffic_bytes_per_launch"] / 1e9))
for r in list(csv.DictReader(open(os.path.join(DST, f"{tag}_bench_prove2p20_kernel_stats.csv"))))[:6]:
    print(r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e6, "ms avg")
