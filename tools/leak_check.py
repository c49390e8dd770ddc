import importlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(16)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
pv = dvp.proving.Prover(inst)
pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
ref = pv.prove(pub, prv)
free0 = None
for it in range(120):
    dvp.set_devices([[0, 0], [0, 0, 0], []][it % 3])
    assert pv.prove(pub, prv) == ref
    if it % 3 == 2:
        torch.cuda.synchronize()
        f = torch.cuda.mem_get_info()[0]
        if free0 is None: free0 = f
        if it % 30 == 29: print(it, "free GB %.3f (delta %.1f MB)" % (f / 1e9, (f - free0) / 1e6), flush=True)
dvp.set_devices([])
# range-keyed rebuilds of the fixed-base tables
be = dvp.distributed.GpuBackend(pv, torch.device("cuda", 0))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
be.begin(w, True)
f0 = torch.cuda.mem_get_info()[0]
for it in range(60):
    lo = (it * 977) % 60000
    be.msm_partial(0, lo, lo + 70000)
torch.cuda.synchronize()
print("after 60 table rebuilds: delta %.1f MB" % ((torch.cuda.mem_get_info()[0] - f0) / 1e6))
pv.close()
