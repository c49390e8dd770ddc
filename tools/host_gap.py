"""Wall time per proof against the GPU-side span of the same proofs (HIP events around dvp_prove_dev's device work,
profile slot prove_total): the difference is host time the GPU spends idle between and inside proofs."""
import ctypes as C, importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_m = int(os.environ.get("LOG_M", "20"))
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
pv = dvp.proving.Prover(inst)
pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
for _ in range(3):
    pv.prove_dev(w.data_ptr(), 0)
dvp.lib.dvp_profile_enable(1)
dvp.lib.dvp_profile_reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = int(os.environ.get("N_PROOFS", "20"))
for _ in range(N):
    pv.prove_dev(w.data_ptr(), 0)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / N * 1e3
ms, n = C.c_double(0), C.c_uint64(0)
out = {}
for name in ("prove_total", "msm_total", "extend_total", "msm_sort", "msm_affine_round0", "msm_affine_rest", "msm_tail"):
    dvp.check(dvp.lib.dvp_profile_read(name.encode(), C.byref(ms), C.byref(n)))
    out[name] = ms.value / N
dvp.lib.dvp_profile_enable(0)
print("wall per proof %.3f ms; GPU-side spans per proof:" % wall, {k: round(v, 3) for k, v in out.items()})
print("outside prove_total (finish + host between proofs): %.3f ms" % (wall - out["prove_total"]))
