"""spread of the two-in-flight loop: R repetitions of K proofs (K/2 per prover) in one process: python tools/in_flight_spread.py [K] [R]"""
import importlib, os, sys, threading, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(20)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
provers, srs = [], None
for _ in range(2):
    pv = dvp.proving.Prover(inst)
    srs = srs or dvp.srs.verifier_runs_setup(pv, inst, td)
    pv.set_srs(srs)
    provers.append(pv)
ref = provers[0].prove_dev(w.data_ptr(), 0)
streams = [torch.cuda.Stream() for _ in provers]
def run(k):
    def loop(i):
        for _ in range(k):
            assert provers[i].prove_dev(w.data_ptr(), streams[i].cuda_stream) == ref
    th = [threading.Thread(target=loop, args=(i,)) for i in range(2)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * k) * 1e3
run(2)
print("ms per step:", " ".join(f"{run(K // 2):.2f}" for _ in range(R)), flush=True)
