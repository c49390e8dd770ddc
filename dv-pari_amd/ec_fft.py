"""
Host mirror of the ECFFT interface the reference consumes (ecfft::FFTree through src/ec_fft.rs and
src/proving.rs:410-422).  All arrays are numpy uint64 [..., 4] canonical limbs, or torch CUDA
tensors of the same shape for the *_dev flavours.
"""
import ctypes as C

import numpy as np

from ._native import lib, check, ptr


class FFTree:
    """build_sect_ecfft_tree(domain_len, shift_by_one, base_log_n, minimal) -- src/ec_fft.rs:197-239.
    `minimal` has no meaning here: twiddles are generated lazily per operation."""

    def __init__(self, domain_len: int, shift_by_one: bool = False, base_log_n: int = 0):
        assert domain_len >= 2 and domain_len & (domain_len - 1) == 0
        self.n = domain_len
        self.log_n = domain_len.bit_length() - 1
        h = C.c_void_p()
        check(lib.dvp_ecfft_create(self.log_n, int(shift_by_one), base_log_n, C.byref(h)), "dvp_ecfft_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.dvp_ecfft_destroy(self._h)
            self._h = None

    __del__ = close

    def leaves(self) -> np.ndarray:
        out = np.empty((self.n, 4), dtype=np.uint64)
        check(lib.dvp_ecfft_leaves(self._h, ptr(out)), "dvp_ecfft_leaves")
        return out

    def get_both_domains(self):
        """src/ec_fft.rs:179-189: (D, D') = (even leaves, odd leaves)."""
        l = self.leaves()
        return np.ascontiguousarray(l[0::2]), np.ascontiguousarray(l[1::2])

    def extend(self, evals: np.ndarray) -> np.ndarray:
        """FFTree::extend(evals, Moiety::S1); evals [batch, n/2, 4] or [n/2, 4]."""
        e = np.ascontiguousarray(evals, dtype=np.uint64)
        single = e.ndim == 2
        if single:
            e = e[None]
        assert e.shape[1:] == (self.n // 2, 4), e.shape
        out = np.empty_like(e)
        check(lib.dvp_ecfft_extend(self._h, ptr(e), e.shape[0], ptr(out)), "dvp_ecfft_extend")
        return out[0] if single else out

    def extend_dev(self, d_in: int, batch: int, d_out: int, stream: int = 0):
        check(lib.dvp_ecfft_extend_dev(self._h, d_in, batch, d_out, stream), "dvp_ecfft_extend_dev")

    def enter(self, coeffs: np.ndarray) -> np.ndarray:
        c = np.ascontiguousarray(coeffs, dtype=np.uint64)
        assert c.shape == (self.n, 4)
        out = np.empty_like(c)
        check(lib.dvp_ecfft_enter(self._h, ptr(c), ptr(out)), "dvp_ecfft_enter")
        return out

    def exit(self, evals: np.ndarray) -> np.ndarray:
        e = np.ascontiguousarray(evals, dtype=np.uint64)
        assert e.shape == (self.n, 4)
        out = np.empty_like(e)
        check(lib.dvp_ecfft_exit(self._h, ptr(e), ptr(out)), "dvp_ecfft_exit")
        return out

    def enter_dev(self, d_in: int, d_out: int, stream: int = 0):
        check(lib.dvp_ecfft_enter_dev(self._h, d_in, d_out, stream), "dvp_ecfft_enter_dev")

    def exit_dev(self, d_in: int, d_out: int, stream: int = 0):
        check(lib.dvp_ecfft_exit_dev(self._h, d_in, d_out, stream), "dvp_ecfft_exit_dev")

    # ---- setup-side operations -------------------------------------------------------
    def vanish_at(self, which: int, x: int) -> int:
        """Z_D(x) (which = 0, D = even leaves) or Z_D'(x) (which = 1) through the isogeny chain: the value
        DensePolynomial::evaluate gives on z_poly / z_polyd (src/ec_fft.rs:362, 424-437)"""
        from . import fr

        out = np.zeros(4, dtype=np.uint64)
        check(lib.dvp_ecfft_vanish_at(self._h, which, ptr(fr.limbs(x)), ptr(out)), "dvp_ecfft_vanish_at")
        return fr.to_int(out)

    def domain_tables(self, which: int = 0):
        """On a 2m-leaf tree (TREE_2N): (bar_wts, z_vals2inv) for which = 0, (bar_wtsd, z_vals2dinv) for which = 1
        -- compute_barycentric_weights + evaluate_vanishing_poly_at_domain/batch_inversion, src/ec_fft.rs:284-335,
        407-419, src/srs.rs:267-311 -- straight from the isogeny chain, no vanishing-polynomial coefficients needed."""
        m = self.n // 2
        a = np.zeros((m, 4), dtype=np.uint64)
        b = np.zeros((m, 4), dtype=np.uint64)
        check(lib.dvp_ecfft_domain_tables(self._h, which, ptr(a), ptr(b)), "dvp_ecfft_domain_tables")
        return a, b


def compute_vanishing_polynomial(tree2n: FFTree, which: int = 0) -> np.ndarray:
    """compute_vanishing_polynomial, src/ec_fft.rs:241-282: the m+1 coefficients of Z_D (which = 0) or Z_D'
    (which = 1), m = leaves/2.  Z on the 2m leaves (zero on its own half, 1/z_vals2inv on the other) -> FFTree::exit
    on the GPU -> truncate; the upper m-1 coefficients must vanish and the polynomial must be monic."""
    from . import fr

    m = tree2n.n // 2
    _, zinv = tree2n.domain_tables(which)
    ev = np.zeros((2 * m, 4), dtype=np.uint64)
    ev[(1 - which)::2] = fr.batch_inverse(zinv)
    co = tree2n.exit(ev)
    if co[m + 1:].any() or fr.to_int(co[m]) != 1:
        raise ArithmeticError("vanishing polynomial is not monic of degree m")
    return np.ascontiguousarray(co[: m + 1])
