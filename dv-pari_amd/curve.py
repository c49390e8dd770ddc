"""Host mirror of src/curve.rs (multi_scalar_mul, point_scalar_mul_gen, to_bytes/from_bytes).

Points are numpy uint64 [n, 8] = x||y (4 limbs each) of the prime-order representative on
y^2+xy=x^3+1, plus an optional uint8 [n] infinity mask; scalars are uint64 [n, 4] canonical."""
import ctypes as C

import numpy as np

from ._native import lib, check, ptr

FR_MODULUS = 3450873173395281893717377931138512760570940988862252126328087024741343


def multi_scalar_mul(scalars: np.ndarray, points_xy: np.ndarray, points_inf: np.ndarray = None):
    """multi_scalar_mul(&[Fr], &[CurvePoint]) -> CurvePoint, src/curve.rs:141-158.
    Returns (xy[8] uint64, is_infinity)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    b = np.ascontiguousarray(points_xy, dtype=np.uint64)
    n = s.shape[0]
    assert s.shape == (n, 4) and b.shape == (n, 8), (s.shape, b.shape)
    inf_p = None
    if points_inf is not None:
        pi = np.ascontiguousarray(points_inf, dtype=np.uint8)
        assert pi.shape == (n,)
        inf_p = ptr(pi)
    out = np.zeros(8, dtype=np.uint64)
    is_inf = C.c_int(0)
    check(lib.dvp_msm_affine(ptr(s), ptr(b), inf_p, n, ptr(out), C.byref(is_inf)), "dvp_msm_affine")
    return out, bool(is_inf.value)


def multi_scalar_mul_dev(d_scalars: int, d_bases: int, d_inf: int, n: int, d_out_xy: int, d_out_inf: int, stream: int = 0):
    check(lib.dvp_msm_affine_dev(d_scalars, d_bases, d_inf, n, d_out_xy, d_out_inf, stream), "dvp_msm_affine_dev")
