"""A/B sweep of run-time tuning knobs on the 2^20 prove: python tools/tune_sweep.py KNOB=v1,v2,... [KNOB2=...]  (cartesian)"""
import importlib, itertools, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_m = int(os.environ.get("LOG_M", "20"))
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
knobs = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in sys.argv[1:]]
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
ref = None
for combo in itertools.product(*[v for _, v in knobs]):
    kw = {k: v for (k, _), v in zip(knobs, combo)}
    with dvp.tune(**kw):
        pv = dvp.proving.Prover(inst)
        pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
        for _ in range(2):
            p = pv.prove_dev(w.data_ptr(), 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            p = pv.prove_dev(w.data_ptr(), 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        if ref is None: ref = p
        assert p == ref
        print(kw, "%.2f ms" % (dt * 1e3), flush=True)
        pv.close()
