"""time of the FIRST proof of a prover (fixed-base tables are built lazily there) against a steady-state proof:
python tools/open_time.py [log_m]   (DVP_LIB=... selects another build)"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
pv0 = dvp.proving.Prover(inst)
srs = dvp.srs.verifier_runs_setup(pv0, inst, td)
pv0.set_srs(srs)
ref = pv0.prove_dev(w.data_ptr(), 0)  # warms everything that is per process (generator table, squaring tables, workspaces)
for rep in range(2):
    t0 = time.perf_counter()
    pv = dvp.proving.Prover(inst)
    pv.set_srs(srs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    p = pv.prove_dev(w.data_ptr(), 0)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    p2 = pv.prove_dev(w.data_ptr(), 0)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    assert p == ref == p2
    print(f"2^{log_m}: create + set_srs {t1 - t0:.3f} s, first proof (builds both tables) {t2 - t1:.3f} s, next proof {(t3 - t2) * 1e3:.1f} ms  [{os.environ.get('DVP_LIB', 'in-tree build')}]", flush=True)
    pv.close()
