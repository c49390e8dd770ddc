"""throughput with P proofs in flight on ONE GPU: P provers (own tables, own stream, own host thread) over the same circuit.
python tools/in_flight.py [log_m] [P ...]   -- every proof is compared with the single-prover bytes"""
import importlib, os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Ps = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
K = 12
STAGGER = float(os.environ.get("STAGGER_MS", "0")) * 1e-3
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
provers, srs = [], None
for _ in range(max(Ps)):
    pv = dvp.proving.Prover(inst)
    srs = srs or dvp.srs.verifier_runs_setup(pv, inst, td)
    pv.set_srs(srs)
    provers.append(pv)
ref = provers[0].prove_dev(w.data_ptr(), 0)
streams = [torch.cuda.Stream() for _ in provers]
for pv, st in zip(provers, streams):
    assert pv.prove_dev(w.data_ptr(), st.cuda_stream) == ref
torch.cuda.synchronize()
import itertools
for slots, P in itertools.product([int(x) for x in os.environ.get("SLOTS", "1,2").split(",")], Ps):
    dvp.lib.dvp_tune_set(b"DVP_MSM_WS_SLOTS", slots)
    bad = []
    def loop(i):
        time.sleep(STAGGER * i)  # STAGGER_MS > 0: staggered arrivals (identical proofs started together stay in lockstep without the gate)
        for _ in range(K):
            if provers[i].prove_dev(w.data_ptr(), streams[i].cuda_stream) != ref:
                bad.append(i)
    warm = [threading.Thread(target=lambda i=i: [provers[i].prove_dev(w.data_ptr(), streams[i].cuda_stream) for _ in range(3)]) for i in range(P)]
    for t in warm: t.start()
    for t in warm: t.join()
    th = [threading.Thread(target=loop, args=(i,)) for i in range(P)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert not bad, bad
    print(f"ws slots {slots}, {P} in flight: {P * K} proofs in {dt * 1e3:.1f} ms = {dt / (P * K) * 1e3:.2f} ms per proof, {(1 << log_m) * P * K / dt / 1e6:.1f} M constraints/s", flush=True)
