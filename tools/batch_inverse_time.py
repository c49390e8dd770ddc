"""k_batch_inverse alone: python tools/batch_inverse_time.py  -> us per call for n = 2^13 (one workgroup: the latency chain) .. 2^21 under every
workgroup shape (DVP_FR_BI_SHAPE)."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
rng = np.random.default_rng(1)
for lg in (13, 17, 21):
    n = 1 << lg
    a = rng.integers(1, 2**62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 38) - 1)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    for shape in (0, 1, 2, 3, 4, 5, 6, 7):
        with dvp.tune(DVP_FR_BI_SHAPE=shape):
            for _ in range(3):
                dvp.check(dvp.lib.dvp_fr_batch_inverse_dev(d.data_ptr(), n, 0))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                dvp.check(dvp.lib.dvp_fr_batch_inverse_dev(d.data_ptr(), n, 0))
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"n=2^{lg} shape {shape}: {dt * 1e6:8.1f} us", flush=True)
