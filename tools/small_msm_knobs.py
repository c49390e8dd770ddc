"""One-shot MSM of 2^lg points under pair-round knobs: python tools/small_msm_knobs.py [lg=16]
(DVP_MSM_AFF_MIN x DVP_MSM_AFF_BMIN x DVP_MSM_BUCKET_PAIRS_MAX; the result must not change)."""
import importlib, itertools, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << lg
rng = np.random.default_rng(3)
def rand(n):
    s = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 39) - 1); return s
xy, inf = dvp.curve.point_scalar_mul_gen_batch(rand(n))
d_s = torch.from_numpy(rand(n).view(np.int64)).cuda(); d_b = torch.from_numpy(xy.view(np.int64)).cuda()
d_out = torch.zeros(8, dtype=torch.int64, device="cuda"); d_inf = torch.zeros(2, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ref = None
for aff_min, bmin, bpm in itertools.product((1 << 19, 1 << 18, 1 << 17, 1 << 16, 1 << 15), (8, 4, 2), (12, 24)):
    with dvp.tune(DVP_MSM_AFF_MIN=aff_min, DVP_MSM_AFF_BMIN=bmin, DVP_MSM_BUCKET_PAIRS_MAX=bpm):
        for _ in range(3):
            dvp.curve.multi_scalar_mul_dev(d_s.data_ptr(), d_b.data_ptr(), 0, n, d_out.data_ptr(), d_inf.data_ptr(), st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            dvp.curve.multi_scalar_mul_dev(d_s.data_ptr(), d_b.data_ptr(), 0, n, d_out.data_ptr(), d_inf.data_ptr(), st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    got = d_out.cpu().numpy().tobytes()
    if ref is None: ref = got
    assert got == ref
    print(f"2^{lg} AFF_MIN=2^{aff_min.bit_length() - 1} BMIN={bmin} BUCKET_PAIRS_MAX={bpm}: {dt * 1e3:.3f} ms", flush=True)
