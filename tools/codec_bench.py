#!/usr/bin/env python3
"""xsk233 encode / decode of a point vector (CurvePoint::to_bytes / from_bytes, the bulk decode of read_point_vec_from_file)."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('', 'tests', 'oracle'):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np
dvp = importlib.import_module("dv-pari_amd")
from util import rand_fr_np
n = 1 << 20
xy, inf = dvp.curve.point_scalar_mul_gen_batch(rand_fr_np(n, 5))
enc = dvp.curve.to_bytes(xy, inf)
for rep in range(3):
    t = time.perf_counter(); xy2, inf2 = dvp.curve.from_bytes(enc); dt = time.perf_counter() - t
    print(f"decode 2^20 incl. H2D/D2H: {dt*1e3:.1f} ms")
assert np.array_equal(xy, xy2) and np.array_equal(inf, inf2)
