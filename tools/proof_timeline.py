#!/usr/bin/env python3
"""Every kernel of ONE proof in launch order out of a rocprofv3 --kernel-trace CSV: start offset, duration, gap before it.
usage: python tools/proof_timeline.py <..._kernel_trace.csv> [proof_index_from_end=1]"""
import csv, sys

def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
    starts = [i for i, r in enumerate(rows) if "k_r1cs_eval" in r[2]]
    lo = starts[-back]
    hi = starts[-back + 1] if back > 1 else len(rows)
    seg = rows[lo:hi]
    t0, prev_end = seg[0][0], seg[0][0]
    for s, e, n in seg:
        name = n.split("(")[0].replace("dvp::", "").replace("void ", "")
        print(f"+{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name[:60]}")
        prev_end = max(prev_end, e)
    print(f"span {(prev_end - t0) / 1e6:.3f} ms, {len(seg)} kernels")

if __name__ == "__main__":
    main()
