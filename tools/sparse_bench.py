#!/usr/bin/env python3
"""Proof::prove on the SP1-like sparse synthetic R1CS (BASELINE config #5 stand-in, dump-format terms, 1..max_terms terms per
side, rows < 2^log_m so the instance is zero-padded).  python tools/sparse_bench.py [log_m] [max_terms]"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
max_terms = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
inst, pub, prv = dvp.gnark_r1cs.synthetic_sparse(log_m, max_terms=max_terms)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
pv = dvp.proving.Prover(inst)
pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
print(f"setup {time.time() - t0:.1f}s rows={inst.n_rows} wires={inst.n_wires} nnz={[int(x.row_ptr[-1]) for x in (inst.l, inst.r, inst.o)]}", flush=True)
dev = torch.device("cuda", 0)
a = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
st = torch.cuda.current_stream().cuda_stream
proof = pv.prove_dev(a.data_ptr(), st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    proof = pv.prove_dev(a.data_ptr(), st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
assert dvp.srs.verify(td, pub, proof)
print(f"sparse 2^{log_m} (<= {max_terms} terms/side): {dt * 1e3:.2f} ms per proof, {(1 << log_m) / dt / 1e6:.1f} M constraints/s (padded size)")
