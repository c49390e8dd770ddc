"""per-launch durations of the steady-state enter and exit between the markers of tools/ecfft_steady.py (rocprofv3 --kernel-trace csv)"""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_batch_inverse" in r["Kernel_Name"]]
assert len(marks) >= 3, marks
for name, (a, b) in (("enter", (marks[-3], marks[-2])), ("exit", (marks[-2], marks[-1]))):
    seg = rows[a + 1:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print(f"== {name}: {len(seg)} launches, span {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
    agg = {}
    for r in seg:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dvp::", "")
        a_ = agg.setdefault(k, [0, 0])
        a_[0] += 1; a_[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:40s} {cnt:4d} {ns / 1e3:9.1f} us  avg {ns / cnt / 1e3:7.1f}")
    with open(os.path.join(d, f"steady_{name}_kernel_stats.csv"), "w", newline="") as fo:  # one steady-state call, no bootstrap launches
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for k, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, cnt, ns, f"{ns / cnt:.1f}", f"{100.0 * ns / busy:.2f}"])
    print("   in order (us):", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in seg))
