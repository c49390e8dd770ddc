"""The MSM shard one rank of an 8-way split runs (1/8 of the 4m-point MSM of a 2^20-constraint proof), 20 times: run it under
rocprofv3 --kernel-trace --stats to see what a SMALL MSM spends its time on (python tools/small_msm_profile.py [parts])"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(20)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
pv = dvp.proving.Prover(inst)
pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
dev = torch.device("cuda", 0)
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
be = dvp.distributed.GpuBackend(pv, dev)
pv2 = dvp.proving.Prover(inst); pv2.set_srs(dvp.srs.verifier_runs_setup(pv2, inst, td))
be2 = dvp.distributed.GpuBackend(pv2, dev)
be2.begin(w, True); full = be2.msm_partial(0, 0, be2.msm_size(0)).clone()
be.begin(w, True); be.challenge(full)
lo, hi = dvp.distributed.shard_range(be.msm_size(1), parts - 1, parts)
for _ in range(3): be.msm_partial(1, lo, hi)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): be.msm_partial(1, lo, hi)
torch.cuda.synchronize()
print("shard of %d pairs: %.3f ms per MSM, plan %s" % (hi - lo, (time.perf_counter() - t0) / 20 * 1e3, pv.msm_plan(1)), flush=True)
