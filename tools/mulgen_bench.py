import importlib, sys, time
import os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
for d in ('', 'tests', 'oracle'): sys.path.insert(0, os.path.join(R, d))
import numpy as np
dvp = importlib.import_module("dv-pari_amd")
from util import rand_fr_np
s = rand_fr_np(1 << 21, 5)
dvp.curve.point_scalar_mul_gen_batch(s[:1000])
for rep in range(3):
    t = time.perf_counter(); xy, inf = dvp.curve.point_scalar_mul_gen_batch(s); dt = time.perf_counter() - t
    print(f"mulgen 2^21 incl. H2D/D2H: {dt*1e3:.1f} ms = {s.shape[0]/dt/1e6:.1f} M/s")
