"""VERDICT r5 item 5, measured before building it: would the [w] half of the commitment MSM, started on a second stream beside
begin (R1CS eval + extends + quotient), shorten a proof?  The commitment MSM is ONE MSM over [w | q2] x [g_m | g_q] today; the probe
times, with HIP events at 2^log_m:
  begin alone; MSM_w alone (fixed-base context over g_m); MSM_q alone (over g_q); the merged MSM (over [g_m | g_q]);
  begin on stream A with MSM_w on stream B (both enqueued together) -- against begin + MSM_w one after the other.
A split only pays if  overlap(begin, MSM_w) + MSM_q  <  begin + merged."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import ctypes as C
dvp = importlib.import_module("dv-pari_amd")
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = 1 << log_m
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
pv = dvp.proving.Prover(inst)
srs = dvp.srs.verifier_runs_setup(pv, inst, td)
pv.set_srs(srs)
dev = torch.device("cuda", 0)
w_host = dvp.fr.vec([1] + pub + prv)
assignment = torch.from_numpy(w_host.view(np.int64)).to(dev)
sA = torch.cuda.current_stream()
sB = torch.cuda.Stream()
proof = pv.prove_dev(assignment.data_ptr(), sA.cuda_stream)
gm, gq = srs.g_m, srs.g_q
ctx_w = dvp.curve.FixedBaseMsm(gm[0], gm[1])
ctx_q = dvp.curve.FixedBaseMsm(gq[0], gq[1])
ctx_all = dvp.curve.FixedBaseMsm(np.concatenate([np.asarray(gm[0]).reshape(-1, 8), np.asarray(gq[0]).reshape(-1, 8)]))
rng = np.random.default_rng(5)
def rand(n):
    s = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 38) - 1); return s
n_w, n_q = ctx_w.n, ctx_q.n
d_sw = assignment  # the witness itself
d_sq = torch.from_numpy(rand(n_q).view(np.int64)).to(dev)
d_sall = torch.cat([assignment, d_sq])
outs = [torch.zeros(10, dtype=torch.int64, device=dev) for _ in range(3)]
def run(ctx, sc, n, out, st):
    dvp.check(dvp.lib.dvp_msm_ctx_run_dev(ctx._h, sc.data_ptr(), 0, n, out.data_ptr(), out.data_ptr() + 64, st.cuda_stream))
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
t_begin = timed(lambda: pv.begin(assignment.data_ptr(), sA.cuda_stream, True))
t_w = timed(lambda: run(ctx_w, d_sw, n_w, outs[0], sA))
t_q = timed(lambda: run(ctx_q, d_sq, n_q, outs[1], sA))
t_all = timed(lambda: run(ctx_all, d_sall, n_w + n_q, outs[2], sA))
def both():
    # the MSM's host side blocks on its largest-bucket read: enqueue begin first, then the MSM from this thread
    pv.begin(assignment.data_ptr(), sA.cuda_stream, True)
    run(ctx_w, d_sw, n_w, outs[0], sB)
t_both = timed(both)
def serial():
    pv.begin(assignment.data_ptr(), sA.cuda_stream, True)
    run(ctx_w, d_sw, n_w, outs[0], sA)
t_serial = timed(serial)
print(f"2^{log_m}: begin {t_begin:.3f} ms | MSM_w ({n_w}) {t_w:.3f} | MSM_q ({n_q}) {t_q:.3f} | merged ({n_w + n_q}) {t_all:.3f} | "
      f"begin then MSM_w {t_serial:.3f} | begin beside MSM_w {t_both:.3f}")
print(f"today: begin + merged = {t_begin + t_all:.3f} ms;  split with overlap: (begin || MSM_w) + MSM_q = {t_both + t_q:.3f} ms;  "
      f"split without overlap: {t_begin + t_w + t_q:.3f} ms")
