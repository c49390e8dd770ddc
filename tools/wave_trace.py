"""Wave-level trace of the batched-affine pair rounds of ONE 2^20 proof (dvp_debug_wave_trace): where the idle wave slots are.

    python tools/wave_trace.py [log_m] [out.json]

Every wave of every k_affine_round launch stamps s_memrealtime (100 MHz) at its start, after pass 1, after the shared inversion
and at its end, s_memtime (shader clock) at start and end, and the hardware slot it ran in.  Per launch this prints / stores:
  * span (first wave start .. last wave end), waves, slots per thread (B), chip-fulls
  * resident-slot fraction = sum of wave lifetimes / (span x 12 slots x CUs seen)           [what SQ_WAVE_CYCLES/BUSY measures]
  * the same split into  ramp (until every CU has its 12 waves), steady, and drain (after the first slot goes idle for good)
  * gap statistics: time a (CU, SIMD, slot) stays empty between one wave's end and its successor's start
  * lifetime spread of the waves (p5 / p50 / p95 / max), per XCD medians, per phase medians (pass 1 / inversion / pass 2)
  * effective shader clock = d(s_memtime) / d(s_memrealtime) per wave (median, p5, p95) if s_memtime runs on the shader clock
"""
import ctypes as C, faulthandler, importlib, json, os, signal, sys
faulthandler.register(signal.SIGUSR1)  # timeout -s USR1 ... prints where a hung run sits
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")

log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(R, "gpurun_out", "wave_trace.json")
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
pv = dvp.proving.Prover(inst)
pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
for _ in range(3):
    ref = pv.prove_dev(w.data_ptr(), 0)
NREC = 1 << 18
buf = torch.zeros(8 + 8 * NREC, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
dvp.check(dvp.lib.dvp_debug_wave_trace(buf.data_ptr(), NREC))
print("tracing one proof", file=sys.stderr, flush=True)
p = pv.prove_dev(w.data_ptr(), 0)
torch.cuda.synchronize()
print("traced", file=sys.stderr, flush=True)
dvp.check(dvp.lib.dvp_debug_wave_trace(None, 0))
assert p == ref or os.environ.get("TRACE_ALLOW_WRONG"), "traced proof differs"
host = buf.cpu().numpy().view(np.uint64)
n = int(host[0])
assert 0 < n <= NREC, n
rec = host[8:8 + 8 * n].reshape(n, 8)
t0, t1, t2, t3, c0, c3 = (rec[:, i].astype(np.int64) for i in range(6))
hw = (rec[:, 6] & 0xffffffff).astype(np.int64)
xcc = ((rec[:, 6] >> 32) & 0xf).astype(np.int64)
blk = (rec[:, 7] & 0xffffffff).astype(np.int64)
tag = ((rec[:, 7] >> 32) & 0xffff).astype(np.int64)
Bs = ((rec[:, 7] >> 48) & 0xffff).astype(np.int64)
wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
slot_key = (cu_key * 4 + simd) * 16 + wave_id
TICK = 1e-5  # ms per s_memrealtime tick (100 MHz)


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else 0.0


report = {"log_m": log_m, "records": n, "launches": []}
for tg in sorted(set(tag.tolist())):
    m = tag == tg
    a0, a1, a2, a3 = t0[m], t1[m], t2[m], t3[m]
    start, end = a0.min(), a3.max()
    span = (end - start) * TICK
    life = (a3 - a0) * TICK
    ncu = len(set(cu_key[m].tolist()))
    nslot = len(set(slot_key[m].tolist()))
    resident = life.sum() / (span * 12 * ncu)
    # per-CU resident waves over time (event sweep), chip-wide
    ev = np.concatenate([np.stack([a0, np.ones_like(a0)], 1), np.stack([a3, -np.ones_like(a3)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    occ = np.cumsum(ev[:, 1])
    full = 12 * ncu
    # ramp: until occupancy first reaches 97 % of full; drain: after it last is >= 97 %
    hi = np.nonzero(occ >= 0.97 * full)[0]
    t_ramp = (ev[hi[0], 0] - start) * TICK if len(hi) else span
    t_drain = (end - ev[hi[-1], 0]) * TICK if len(hi) else 0.0
    # integral of idle slots by stretch
    dt = np.diff(ev[:, 0]) * TICK
    idle = (full - occ[:-1]) * dt
    tt = (ev[:-1, 0] - start) * TICK
    idle_ramp = idle[tt < t_ramp].sum()
    idle_drain = idle[tt >= span - t_drain].sum()
    idle_mid = idle.sum() - idle_ramp - idle_drain
    tot = span * full
    # gaps per hardware slot
    gaps = []
    sk = slot_key[m]
    order = np.lexsort((a0, sk))
    sk_s, s0, s3 = sk[order], a0[order], a3[order]
    same = sk_s[1:] == sk_s[:-1]
    gaps = ((s0[1:] - s3[:-1])[same]) * TICK
    # chip-fulls: waves per slot
    per_slot = np.bincount(np.unique(sk, return_inverse=True)[1])
    clk = (c3[m] - c0[m]) / np.maximum(a3 - a0, 1) * 100.0  # MHz if s_memtime is the shader clock
    xs = xcc[m]
    L = {
        "tag": int(tg), "kernel": "k_affine_round", "waves": int(m.sum()), "slots_per_thread_B": int(np.median(Bs[m])),
        "span_ms": round(span, 4), "cus_seen": ncu, "hw_slots_seen": nslot, "waves_per_slot_median": float(np.median(per_slot)),
        "resident_slot_fraction": round(float(resident), 4),
        "idle_fraction": {"ramp": round(idle_ramp / tot, 4), "steady": round(idle_mid / tot, 4), "drain": round(idle_drain / tot, 4)},
        "ramp_ms": round(float(t_ramp), 4), "drain_ms": round(float(t_drain), 4),
        "gap_us": {"n": int(len(gaps)), "p50": round(pct(gaps, 50) * 1e3, 2), "p95": round(pct(gaps, 95) * 1e3, 2), "max": round(float(gaps.max()) * 1e3 if len(gaps) else 0, 2),
                   "sum_over_slots_fraction": round(float(gaps.sum()) / tot, 4)},
        "wave_life_ms": {"p5": round(pct(life, 5), 4), "p50": round(pct(life, 50), 4), "p95": round(pct(life, 95), 4), "max": round(float(life.max()), 4)},
        "phase_ms_median": {"pass1": round(float(np.median(a1 - a0)) * TICK, 4), "inversion": round(float(np.median(a2 - a1)) * TICK, 4),
                            "pass2": round(float(np.median(a3 - a2)) * TICK, 4)},
        "life_ms_median_by_xcd": {int(x): round(float(np.median(life[xs == x])), 4) for x in sorted(set(xs.tolist()))},
        "waves_by_xcd": {int(x): int((xs == x).sum()) for x in sorted(set(xs.tolist()))},
        "memtime_per_realtime_x100": {"p5": round(pct(clk, 5), 1), "p50": round(pct(clk, 50), 1), "p95": round(pct(clk, 95), 1)},
    }
    report["launches"].append(L)
    if L["waves"] > 2000:
        print(json.dumps(L), flush=True)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(report, open(out_path, "w"), indent=1)
np.save(out_path.replace(".json", ".npy"), rec)
big = max(report["launches"], key=lambda l: l["waves"])
print("largest launch:", json.dumps(big, indent=1))
pv.close()
