"""per-launch digest of a wave trace (tools/wave_trace.py's .npy): life by wave slot, drain, clock.  python tools/wave_trace_digest.py x.npy"""
import sys, numpy as np
rec = np.load(sys.argv[1])
t0, t1, t2, t3, c0, c3 = (rec[:, i].astype(np.int64) for i in range(6))
hw = (rec[:, 6] & 0xffffffff).astype(np.int64)
tag = ((rec[:, 7] >> 32) & 0xffff).astype(np.int64)
B = ((rec[:, 7] >> 48) & 0xffff).astype(np.int64)
wid = hw & 15
T0 = t0.min()
for tg in sorted(set(tag.tolist())):
    m = tag == tg
    st = t0[m].min()
    life = (t3 - t0)[m] * 1e-5
    span = (t3[m].max() - st) * 1e-5
    clk = ((c3 - c0) / np.maximum(t3 - t0, 1))[m] * 100
    first = (t0[m] - st) * 1e-5 < 0.05
    by = {int(v): round(float(np.median(life[first][wid[m][first] == v])), 3) for v in sorted(set(wid[m][first].tolist())) if (wid[m][first] == v).sum() > 50}
    print(f"tag {tg:2d} at {(st - T0) * 1e-5:6.2f} ms  span {span:6.3f} ms  waves {m.sum():5d}  B {int(np.median(B[m])):3d}  life p5/50/95 {np.percentile(life, 5):.3f}/{np.percentile(life, 50):.3f}/{np.percentile(life, 95):.3f}"
          f"  sum(life)/(span*3072) {life.sum() / (span * 3072):.3f}  first-chip-full life by wave slot {by}  clk p50 {np.median(clk):.0f} MHz")
