"""quad-cooperative reducer threshold sweep on small one-shot MSMs and on the fixed-base shard of an 8-way split"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
def rand_scalars(n, seed):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 39) - 1)
    return s
st = torch.cuda.current_stream().cuda_stream
for ln in (16, 17):
    n = 1 << ln
    xy, inf = dvp.curve.point_scalar_mul_gen_batch(rand_scalars(n, 1))
    d_s = torch.from_numpy(rand_scalars(n, 2).view(np.int64)).cuda(); d_b = torch.from_numpy(xy.view(np.int64)).cuda()
    d_out = torch.zeros(8, dtype=torch.int64, device="cuda"); d_inf = torch.zeros(2, dtype=torch.int32, device="cuda")
    def run(reps):
        for _ in range(reps):
            dvp.curve.multi_scalar_mul_dev(d_s.data_ptr(), d_b.data_ptr(), 0, n, d_out.data_ptr(), d_inf.data_ptr(), st)
        torch.cuda.synchronize()
    run(2); ref = d_out.clone()
    for qm in (1, 49152, 98304, 196608, 393216):
        for K in (4, 8):
            with dvp.tune(DVP_MSM_ACCUM_QUAD_MAX=qm, DVP_MSM_K=K):
                run(2); assert (ref == d_out).all()
                t0 = time.perf_counter(); run(10); dt = (time.perf_counter() - t0) / 10 * 1e3
            print(f"one-shot 2^{ln} accum_quad_max={qm} K={K}: {dt:.3f} ms", flush=True)
# the K-MSM shard of the last rank of an 8-way split (fixed-base tables for that slice)
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(20)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
pv = dvp.proving.Prover(inst); pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
dev = torch.device("cuda", 0)
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
be = dvp.distributed.GpuBackend(pv, dev)
be.begin(w, True); full = be.msm_partial(0, 0, be.msm_size(0)).clone(); be.challenge(full)
lo, hi = dvp.distributed.shard_range(be.msm_size(1), 7, 8)
for qm in (1, 49152, 98304, 196608, 393216, 786432):
    for K in (4, 8):
        with dvp.tune(DVP_MSM_ACCUM_QUAD_MAX=qm, DVP_MSM_K=K):
            for _ in range(3): be.msm_partial(1, lo, hi)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): be.msm_partial(1, lo, hi)
            torch.cuda.synchronize()
            print(f"shard of {hi - lo} pairs accum_quad_max={qm} K={K}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms plan {pv.msm_plan(1)}", flush=True)
