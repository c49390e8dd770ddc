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


def point_scalar_mul_gen_batch(scalars: np.ndarray):
    """point_scalar_mul_gen over a vector (src/curve.rs:129-137, loops at src/srs.rs:130-160).
    Returns (xy [n,8], inf [n])."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = s.shape[0]
    xy = np.zeros((n, 8), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    check(lib.dvp_mulgen_batch_affine(ptr(s), n, ptr(xy), ptr(inf)), "dvp_mulgen_batch_affine")
    return xy, inf


def point_scalar_mul_gen_batch_bytes(scalars: np.ndarray) -> np.ndarray:
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = s.shape[0]
    out = np.zeros((n, 30), dtype=np.uint8)
    check(lib.dvp_mulgen_batch(ptr(s), n, ptr(out)), "dvp_mulgen_batch")
    return out


def to_bytes(points_xy: np.ndarray, points_inf: np.ndarray = None) -> np.ndarray:
    """CurvePoint::to_bytes over a vector, src/curve.rs:93-100."""
    b = np.ascontiguousarray(points_xy, dtype=np.uint64).reshape(-1, 8)
    n = b.shape[0]
    inf_p = None
    if points_inf is not None:
        pi = np.ascontiguousarray(points_inf, dtype=np.uint8)
        inf_p = ptr(pi)
    out = np.zeros((n, 30), dtype=np.uint8)
    check(lib.dvp_points_encode(ptr(b), inf_p, n, ptr(out)), "dvp_points_encode")
    return out


def from_bytes(enc: np.ndarray):
    """CurvePoint::from_bytes over a vector, src/curve.rs:103-109; raises DvpError(DVP_EDECODE)
    on an invalid encoding (the reference asserts, src/io_utils.rs:223)."""
    e = np.ascontiguousarray(enc, dtype=np.uint8).reshape(-1, 30)
    n = e.shape[0]
    xy = np.zeros((n, 8), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    check(lib.dvp_points_decode(ptr(e), n, ptr(xy), ptr(inf)), "dvp_points_decode")
    return xy, inf


def add(a_xy, b_xy, a_inf=None, b_inf=None):
    """CurvePoint::add over two vectors, src/curve.rs:84-90.  Returns (xy [n,8], inf [n])."""
    a = np.ascontiguousarray(a_xy, dtype=np.uint64).reshape(-1, 8)
    b = np.ascontiguousarray(b_xy, dtype=np.uint64).reshape(-1, 8)
    n = a.shape[0]
    assert b.shape[0] == n
    ai = None if a_inf is None else np.ascontiguousarray(a_inf, dtype=np.uint8)
    bi = None if b_inf is None else np.ascontiguousarray(b_inf, dtype=np.uint8)
    xy = np.zeros((n, 8), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    check(lib.dvp_points_add(ptr(a), None if ai is None else ptr(ai), ptr(b), None if bi is None else ptr(bi), n, ptr(xy), ptr(inf)),
          "dvp_points_add")
    return xy, inf


def multi_scalar_mul_bytes(scalars32: np.ndarray, bases30: np.ndarray) -> bytes:
    s = np.ascontiguousarray(scalars32, dtype=np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(bases30, dtype=np.uint8).reshape(-1, 30)
    assert s.shape[0] == b.shape[0]
    out = np.zeros(30, dtype=np.uint8)
    check(lib.dvp_msm_xsk233(ptr(s), ptr(b), s.shape[0], ptr(out)), "dvp_msm_xsk233")
    return out.tobytes()


class FixedBaseMsm:
    """multi_scalar_mul against a fixed base vector (an SRS vector): dvp_msm_ctx_* in include/dvpari.h."""

    def __init__(self, points_xy: np.ndarray, points_inf: np.ndarray = None, range_hint: int = 0):
        b = np.ascontiguousarray(points_xy, dtype=np.uint64).reshape(-1, 8)
        self.n = b.shape[0]
        inf_p = None
        if points_inf is not None:
            pi = np.ascontiguousarray(points_inf, dtype=np.uint8)
            inf_p = ptr(pi)
        h = C.c_void_p()
        check(lib.dvp_msm_ctx_create(ptr(b), inf_p, self.n, range_hint, C.byref(h)), "dvp_msm_ctx_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and lib is not None:  # lib is None while the interpreter shuts down
            lib.dvp_msm_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def plan(self):
        """(window bits, windows) the context settled on"""
        c, w = C.c_int(0), C.c_int(0)
        check(lib.dvp_msm_ctx_plan(self._h, C.byref(c), C.byref(w)), "dvp_msm_ctx_plan")
        return c.value, w.value

    def table(self):
        """(bytes of HBM, signed binary windows -- the default flavour -- ?) of the pre-rotated table"""
        s = C.c_int(0)
        b = int(lib.dvp_msm_ctx_table_bytes(self._h, C.byref(s)))
        return b, bool(s.value)

    def run(self, scalars: np.ndarray, lo: int = 0, hi: int = None):
        hi = self.n if hi is None else hi
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        assert s.shape[0] == hi - lo
        out = np.zeros(8, dtype=np.uint64)
        is_inf = C.c_int(0)
        check(lib.dvp_msm_ctx_run(self._h, ptr(s), lo, hi, ptr(out), C.byref(is_inf)), "dvp_msm_ctx_run")
        return out, bool(is_inf.value)
