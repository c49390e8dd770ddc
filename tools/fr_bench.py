"""Fr vector kernels alone: batch inversion of 2^21 elements (the [den | den2] inversion of a 2^20-constraint proof) -- python tools/fr_bench.py"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
from util import rand_fr_np
n = 1 << 21
a = rand_fr_np(n, 5)
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
