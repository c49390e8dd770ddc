"""a few fixed-base MSMs of 2^lg points (default plan) for rocprofv3 --kernel-trace: python tools/fixed_msm_trace.py [lg]"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << lg
rng = np.random.default_rng(7)
def rand_fr(n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); a[:, 3] &= (1 << 38) - 1
    return a
bases, _ = dvp.curve.point_scalar_mul_gen_batch(rand_fr(n))
fb = dvp.curve.FixedBaseMsm(bases)
d_s = torch.from_numpy(rand_fr(n).view(np.int64)).cuda()
out = torch.zeros(10, dtype=torch.int64, device="cuda")
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dvp.check(dvp.lib.dvp_msm_ctx_run_dev(fb._h, d_s.data_ptr(), 0, n, out.data_ptr(), out.data_ptr() + 64, 0), "run")
    torch.cuda.synchronize()
    print(f"2^{lg} plan {fb.plan()} run {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
