"""Fr vectors as numpy uint64 [n,4] canonical limbs, with the GPU vector ops of the C ABI.
(src/curve.rs:16-22; the rayon pointwise maps of src/proving.rs:492-654.)"""
import numpy as np

from ._native import lib, check, ptr

P = 3450873173395281893717377931138512760570940988862252126328087024741343
MASK64 = (1 << 64) - 1


def limbs(v) -> np.ndarray:
    """python int -> [4] uint64"""
    v = int(v) % P
    return np.array([(v >> (64 * i)) & MASK64 for i in range(4)], dtype=np.uint64)


def vec(vals) -> np.ndarray:
    vals = [int(v) % P for v in vals]
    buf = b"".join(v.to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u8").reshape(len(vals), 4).copy()


def to_int(a) -> int:
    return int.from_bytes(np.ascontiguousarray(a, dtype="<u8").tobytes(), "little")


def to_ints(arr) -> list:
    raw = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, 4).tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


def mul(a, b):
    a, b = _c(a), _c(b)
    out = np.empty_like(a)
    check(lib.dvp_fr_vec_mul(ptr(a), ptr(b), a.shape[0], ptr(out)), "dvp_fr_vec_mul")
    return out


def scale(a, s: int):
    a = _c(a)
    out = np.empty_like(a)
    check(lib.dvp_fr_vec_scale(ptr(a), ptr(limbs(s)), a.shape[0], ptr(out)), "dvp_fr_vec_scale")
    return out


def axpy(a, s: int, b):
    """a + s * b"""
    a, b = _c(a), _c(b)
    out = np.empty_like(a)
    check(lib.dvp_fr_vec_axpy(ptr(a), ptr(limbs(s)), ptr(b), a.shape[0], ptr(out)), "dvp_fr_vec_axpy")
    return out


def scalar_sub(s: int, a):
    a = _c(a)
    out = np.empty_like(a)
    check(lib.dvp_fr_vec_scalar_sub(ptr(limbs(s)), ptr(a), a.shape[0], ptr(out)), "dvp_fr_vec_scalar_sub")
    return out


def dot(a, b) -> int:
    a, b = _c(a), _c(b)
    out = np.zeros(4, dtype=np.uint64)
    check(lib.dvp_fr_vec_dot(ptr(a), ptr(b), a.shape[0], ptr(out)), "dvp_fr_vec_dot")
    return to_int(out)


def batch_inverse(a):
    """ark_ff::batch_inversion (zeros stay zero)."""
    a = _c(a).copy()
    check(lib.dvp_fr_batch_inverse(ptr(a), a.shape[0]), "dvp_fr_batch_inverse")
    return a


def spmv(row_ptr, col, coeff_ids, coeffs, x):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    coeff_ids = np.ascontiguousarray(coeff_ids, dtype=np.uint32)
    coeffs, x = _c(coeffs), _c(x)
    n_rows = row_ptr.shape[0] - 1
    out = np.zeros((n_rows, 4), dtype=np.uint64)
    check(lib.dvp_fr_spmv(ptr(row_ptr), ptr(col), ptr(coeff_ids), n_rows, ptr(coeffs), coeffs.shape[0], ptr(x), x.shape[0], ptr(out)),
          "dvp_fr_spmv")
    return out
