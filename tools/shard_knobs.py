"""knob sweep on the K-MSM shard of the last rank of a W-way split (fixed-base tables of that slice): python tools/shard_knobs.py [W]"""
import importlib, itertools, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(20)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
pv = dvp.proving.Prover(inst); pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
dev = torch.device("cuda", 0)
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
be = dvp.distributed.GpuBackend(pv, dev)
be.begin(w, True); full = be.msm_partial(0, 0, be.msm_size(0)).clone(); be.challenge(full)
lo, hi = dvp.distributed.shard_range(be.msm_size(1), W - 1, W)
ref = None
for am, qm in itertools.product((1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20), (16384, 32768, 65536)):
    with dvp.tune(DVP_MSM_AFF_MIN=am, DVP_MSM_QUAD_MAX=qm):
        for _ in range(3): out = be.msm_partial(1, lo, hi)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        assert torch.equal(ref, out)
        t0 = time.perf_counter()
        for _ in range(20): be.msm_partial(1, lo, hi)
        torch.cuda.synchronize()
        print(f"W={W} shard of {hi - lo} pairs aff_min=2^{am.bit_length() - 1} quad_max={qm}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms plan {pv.msm_plan(1)}", flush=True)
