#!/usr/bin/env python3
"""Timeline of ONE proof out of a rocprofv3 --kernel-trace CSV: where the stream sits idle between kernels.

usage: python tools/timeline_gaps.py <..._kernel_trace.csv> [proof_index_from_end=1] [top=25]

A proof starts at its k_r1cs_eval launch (the first kernel of dvp_prove_dev) and ends with the kernel before the next one
(or the last kernel of the trace).  Prints span, busy time, idle time, the per-kernel-name sums inside the proof and the
largest gaps with the kernels either side of them (a gap after a kernel the host waits on is a host round trip)."""
import csv, sys, collections

def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "k_r1cs_eval" in r[2]]
    if not starts:
        sys.exit("no k_r1cs_eval in the trace")
    lo = starts[-back]
    hi = starts[-back + 1] if back > 1 else len(rows)
    seg = rows[lo:hi]
    t0 = seg[0][0]
    span = (max(r[1] for r in seg) - t0) / 1e6
    busy, cur_end, gaps = 0.0, seg[0][0], []
    for i, (s, e, n) in enumerate(seg):
        if s > cur_end:
            gaps.append(((s - cur_end) / 1e3, i))
            busy += (e - s) / 1e6
            cur_end = e
        else:
            if e > cur_end:
                busy += (e - cur_end) / 1e6
                cur_end = e
    print(f"proof {back} from the end: {len(seg)} kernels, span {span:.3f} ms, busy {busy:.3f} ms, idle {span - busy:.3f} ms")
    by = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in seg:
        k = n.split("(")[0][-60:]
        by[k][0] += 1
        by[k][1] += (e - s) / 1e6
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"  {t:8.3f} ms  x{c:<4d} {k}")
    print("largest gaps (us): after -> before")
    for g, i in sorted(gaps, reverse=True)[:top]:
        print(f"  {g:8.1f}  at +{(seg[i][0] - t0) / 1e6:7.3f} ms  {seg[i - 1][2].split('(')[0][-40:]} -> {seg[i][2].split('(')[0][-40:]}")
    small = sum(g for g, _ in gaps if g < 20)
    print(f"gaps: {len(gaps)} totalling {sum(g for g, _ in gaps) / 1e3:.3f} ms; of which < 20 us each: {small / 1e3:.3f} ms")

if __name__ == "__main__":
    main()
