"""Soak test of concurrent provers on one GPU: one host thread + stream per prover (sizes 2^12, 2^16, 2^20, 2^20), proofs back to
back for the given time, every one byte-compared with the bytes its prover computed alone (catches ordering bugs between the two
MSM workspaces of a device, the pair-round gate and the provers' own streams).  python tools/soak_in_flight.py [minutes]"""
import importlib, os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120.0
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
cases = []
for log_m in (12, 16, 20, 20):
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
    ref = pv.prove_dev(w.data_ptr(), 0)
    assert dvp.srs.verify(td, pub, ref)
    cases.append((log_m, pv, w, ref, torch.cuda.Stream()))
t0 = time.time()
count = [0] * len(cases)
bad = []
def loop(i):
    log_m, pv, w, ref, st = cases[i]
    while time.time() - t0 < budget and not bad:
        if pv.prove_dev(w.data_ptr(), st.cuda_stream) != ref:
            bad.append((i, log_m, count[i]))
        count[i] += 1
th = [threading.Thread(target=loop, args=(i,)) for i in range(len(cases))]
for t in th: t.start()
while any(t.is_alive() for t in th):
    time.sleep(30)
    print(f"{sum(count)} proofs ok ({count}), {time.time() - t0:.0f}s", flush=True)
for t in th: t.join()
assert not bad, bad
free, total = torch.cuda.mem_get_info()
print(f"done: {sum(count)} proofs {count}, all identical; device memory in use {(total - free) / 1e9:.1f} GB")
