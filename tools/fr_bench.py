"""Fr vector kernels alone: batch inversion of 2^21 elements (the [den | den2] inversion of a 2^20-constraint proof) -- python tools/fr_bench.py"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
n = 1 << 21
rng = np.random.default_rng(5)
a = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
a[:, 3] &= np.uint64((1 << 38) - 1)  # < 2^230 < p
a[17] = 0
d = torch.from_numpy(a.view(np.int64)).cuda()
ref = d.clone()
for _ in range(2):
    dvp.check(dvp.lib.dvp_fr_batch_inverse_dev(d.data_ptr(), n, 0))  # inverse twice = identity
torch.cuda.synchronize()
assert torch.equal(d, ref), "inverse o inverse != id"
t0 = time.perf_counter()
for _ in range(20):
    dvp.check(dvp.lib.dvp_fr_batch_inverse_dev(d.data_ptr(), n, 0))
torch.cuda.synchronize()
print("batch inverse of 2^21: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
