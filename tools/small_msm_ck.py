"""one-shot MSM of 2^lg points under DVP_MSM_C x DVP_MSM_K (window bits x reducer fan-in): has the optimum moved?"""
import sys, importlib, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd"); nat = importlib.import_module("dv-pari_amd._native")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << lg
rng = np.random.default_rng(1)
def rs(n):
    s = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 39) - 1); return s
xy, inf = dvp.curve.point_scalar_mul_gen_batch(rs(n))
d_s = torch.from_numpy(rs(n).view(np.int64)).cuda(); d_b = torch.from_numpy(xy.view(np.int64)).cuda()
d_out = torch.zeros(8, dtype=torch.int64, device="cuda"); d_inf = torch.zeros(2, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def t():
    for _ in range(2): dvp.curve.multi_scalar_mul_dev(d_s.data_ptr(), d_b.data_ptr(), 0, n, d_out.data_ptr(), d_inf.data_ptr(), st)
    torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); dvp.curve.multi_scalar_mul_dev(d_s.data_ptr(), d_b.data_ptr(), 0, n, d_out.data_ptr(), d_inf.data_ptr(), st); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
ref = None
print(f"2^{lg}: default {t():.3f} ms")
for c in (8, 9, 10, 11, 12, 13):
    row = []
    for k in (4, 6, 8, 12, 16):
        with nat.tune(DVP_MSM_C=c, DVP_MSM_K=k):
            ms = t(); cur = d_out.clone()
        if ref is None: ref = cur
        assert torch.equal(cur, ref), (c, k)
        row.append(f"K={k}: {ms:.3f}")
    print(f"  c={c}: " + "  ".join(row), flush=True)
