#!/bin/bash
# Runs ON THE MI355X BOX: kernel trace of one-shot MSMs of 2^$1 points (tools/msm_bench.py); prints the launches of the LAST MSM in
# stream order with their durations and the gaps in front of them.   -> gpurun_out/msm_trace/
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
LG=${1:-16}
OUT=$ROOT/gpurun_out/msm_trace
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o m --output-format csv -- python3 $ROOT/tools/msm_bench.py $LG > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# an MSM ends with its k_tail launch: take the launches between the last two
tails = [i for i, r in enumerate(rows) if "k_tail<" in r["Kernel_Name"]]
seg = rows[tails[-2] + 1:tails[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"]); prev = t0
print(f"{len(seg)} launches, span {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, busy {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3:.1f} us")
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dvp::", "")
    print(f"  +{(s - t0) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?'):>9}  {name}")
    prev = e
PY
