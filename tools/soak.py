"""Soak test: many proofs in a row, alternating sizes / device lists, every one byte-compared with the first (catches rare
ordering bugs between the MSM's side stream, the per-device workspaces and the prover's own stream).  python tools/soak.py [minutes]"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120.0
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
cases = []
for log_m in (12, 16, 20):
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
    ref = pv.prove_dev(w.data_ptr(), 0)
    assert dvp.srs.verify(td, pub, ref)
    cases.append((log_m, pv, w, ref))
t0 = time.time(); n = 0
streams = [torch.cuda.Stream() for _ in range(2)]
while time.time() - t0 < budget:
    for log_m, pv, w, ref in cases:
        devs = [[], [0, 0], [0, 0, 0]][n % 3]
        dvp.set_devices(devs)
        st = streams[n % 2]
        with torch.cuda.stream(st):
            p = pv.prove_dev(w.data_ptr(), st.cuda_stream)
        assert p == ref, (n, log_m, devs)
        n += 1
    if n % 60 == 0:
        print(f"{n} proofs ok, {time.time() - t0:.0f}s", flush=True)
dvp.set_devices([])
print(f"soak done: {n} proofs, all identical", flush=True)
