"""Knob sweep for SMALL one-shot MSMs (config #2: 2^16 points; also what a rank of an 8-way split sees): window size x pair-round
threshold x reducer fan-in.  python tools/small_msm_sweep.py [log_n ..]"""
import importlib, itertools, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
def rand_scalars(n, seed):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 39) - 1)
    return s
st = torch.cuda.current_stream().cuda_stream
for ln in [int(x) for x in sys.argv[1:]] or [16]:
    n = 1 << ln
    xy, inf = dvp.curve.point_scalar_mul_gen_batch(rand_scalars(n, 1))
    d_s = torch.from_numpy(rand_scalars(n, 2).view(np.int64)).cuda(); d_b = torch.from_numpy(xy.view(np.int64)).cuda()
    d_out = torch.zeros(8, dtype=torch.int64, device="cuda"); d_inf = torch.zeros(2, dtype=torch.int32, device="cuda")
    def run(reps):
        for _ in range(reps):
            dvp.curve.multi_scalar_mul_dev(d_s.data_ptr(), d_b.data_ptr(), 0, n, d_out.data_ptr(), d_inf.data_ptr(), st)
        torch.cuda.synchronize()
    run(2); ref = d_out.clone()
    t0 = time.perf_counter(); run(10); base = (time.perf_counter() - t0) / 10 * 1e3
    print(f"n=2^{ln} defaults: {base:.3f} ms", flush=True)
    res = []
    cs = [int(x) for x in os.environ.get("CS", "10,11,12,13").split(",")]
    for c, am, K in itertools.product(cs, (1 << 13, 1 << 15, 1 << 16, 1 << 17, 1 << 19), (4, 8)):
        with dvp.tune(DVP_MSM_C=c, DVP_MSM_AFF_MIN=am, DVP_MSM_K=K):
            run(2); assert (ref == d_out).all()
            t0 = time.perf_counter(); run(8); dt = (time.perf_counter() - t0) / 8 * 1e3
        res.append((dt, c, am, K))
    for dt, c, am, K in sorted(res)[:12]:
        print(f"  c={c} aff_min=2^{am.bit_length() - 1} K={K}: {dt:.3f} ms ({n / dt / 1e3:.1f} Mpoints/s)", flush=True)
