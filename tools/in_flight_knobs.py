"""two proofs in flight (tools/in_flight.py's measurement) under knob settings: python tools/in_flight_knobs.py KNOB=v1,v2 ..
(the knobs that trade latency for work may have different optima when a second proof hides the latency)"""
import importlib, itertools, os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
sweep = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in sys.argv[1:]]
K, P = 12, 2
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(20)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
provers, srs = [], None
for _ in range(P):
    pv = dvp.proving.Prover(inst)
    srs = srs or dvp.srs.verifier_runs_setup(pv, inst, td)
    pv.set_srs(srs)
    provers.append(pv)
ref = provers[0].prove_dev(w.data_ptr(), 0)
streams = [torch.cuda.Stream() for _ in provers]
def run(k):
    bad = []
    def loop(i):
        for _ in range(k):
            if provers[i].prove_dev(w.data_ptr(), streams[i].cuda_stream) != ref: bad.append(i)
    th = [threading.Thread(target=loop, args=(i,)) for i in range(P)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert not bad
    return dt
for combo in itertools.product(*[v for _, v in sweep]):
    knobs = {k: v for (k, _), v in zip(sweep, combo)}
    with dvp.tune(**knobs):
        run(3)
        ts = [run(K) / (P * K) * 1e3 for _ in range(3)]
        t1 = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(6): provers[0].prove_dev(w.data_ptr(), 0)
            torch.cuda.synchronize(); t1.append((time.perf_counter() - t0) / 6 * 1e3)
    print(f"{knobs}: two in flight {sorted(ts)[1]:.2f} ms per proof (min {min(ts):.2f}), alone {sorted(t1)[1]:.2f}", flush=True)
