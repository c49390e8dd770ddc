"""CPU tests (no GPU): the C restatement of the ECFFT oracle (oracle/dvp_oracle_ecfft.c) against the python big-int
oracle (oracle/pyref.py, itself pinned against O(n^2) Lagrange interpolation -- the shape of the reference's own
test_interpolate_and_extend_match, src/ec_fft.rs:883-907) and against the definition of enter (Horner at the leaves).
The C oracle is what the -m gpu tests compare the multi-block ECFFT kernels with at 2^12 .. 2^20."""
import random

import numpy as np
import pytest

import c_oracle as co
import pyref as o
from util import to_limbs, from_limbs, rand_fr_np


@pytest.mark.parametrize("log_n,shifted", [(1, False), (2, False), (3, True), (5, False), (6, True), (8, False)])
def test_c_fftree_equals_pyref(log_n, shifted):
    rnd = random.Random(100 + log_n)
    base = log_n + 2 if shifted else None
    ct = co.FFTree(log_n, shifted, base)
    ot = o.FFTree(log_n, shifted, base)
    for d in range(log_n + 1):
        assert from_limbs(ct.layer(d)) == ot.layers[d]
    n = 1 << log_n
    for sl in range(0, min(log_n, 3)):
        m = (n >> sl) // 2
        ev = [rnd.randrange(o.P) for _ in range(m)]
        for to_even in (False, True):
            assert from_limbs(ct.extend(to_limbs(ev), sl, to_even)) == ot.extend(ev, 1 << sl, to_even)
        c = [rnd.randrange(o.P) for _ in range(2 * m)]
        e = ot.enter(c, 1 << sl)
        assert from_limbs(ct.enter(to_limbs(c), sl)) == e
        assert from_limbs(ct.eval(to_limbs(c), sl)) == e
        assert from_limbs(ct.exit(to_limbs(e), sl)) == c
        ev2 = [rnd.randrange(o.P) for _ in range(2 * m)]
        assert from_limbs(ct.exit(to_limbs(ev2), sl)) == ot.exit(ev2, 1 << sl)
    x = rnd.randrange(o.P)
    assert ct.vanish_even_at(x) == ot.vanish_even_at(x)


def test_c_fftree_extend_is_interpolation():
    """extend == interpolate on D, evaluate on D' (src/ec_fft.rs:883-907), here against python Lagrange at n = 32"""
    rnd = random.Random(5)
    ct = co.FFTree(6)
    leaves = from_limbs(ct.leaves())
    d, d2 = leaves[0::2], leaves[1::2]
    ev = [rnd.randrange(o.P) for _ in range(32)]
    out = from_limbs(ct.extend(to_limbs(ev)))
    assert out == [o.lagrange_eval(d, ev, x) for x in d2]


def test_c_fftree_matrices_are_lemma_3_2():
    """decompose = inverse of recombine-shaped matrix on the source pairs; recombine on the destination pairs"""
    log_n = 5
    ct = co.FFTree(log_n)
    ot = o.FFTree(log_n)
    n = (1 << log_n) // 2
    for to_even in (False, True):
        dec = from_limbs(ct.matrices(to_even, 0))
        rec = from_limbs(ct.matrices(to_even, 1))
        src, dst = (1, 0) if to_even else (0, 1)
        for d in range(log_n - 1):
            nd, off = n >> d, n - (n >> d)
            L, x0, e = ot.layers[d], ot.x0[d], (nd >> 1) - 1
            for i in range(nd >> 1):
                s0, s1 = L[2 * i + dst], L[2 * i + dst + nd]
                v0, v1 = pow(s0 - x0, e, o.P), pow(s1 - x0, e, o.P)
                assert rec[4 * (off + i):4 * (off + i) + 4] == [v0, s0 * v0 % o.P, v1, s1 * v1 % o.P]
                s0, s1 = L[2 * i + src], L[2 * i + src + nd]
                v0, v1 = pow(s0 - x0, e, o.P), pow(s1 - x0, e, o.P)
                m = [v0, s0 * v0 % o.P, v1, s1 * v1 % o.P]
                a, b, c, dd = dec[4 * (off + i):4 * (off + i) + 4]
                # dec * m == identity
                assert [(a * m[0] + b * m[2]) % o.P, (a * m[1] + b * m[3]) % o.P, (c * m[0] + dd * m[2]) % o.P, (c * m[1] + dd * m[3]) % o.P] == [1, 0, 0, 1]


def test_c_fftree_mid_size_self_consistency():
    """2^11 leaves (beyond what pyref does in a test run): enter == Horner, exit o enter == id, extend links D and D'"""
    log_n = 11
    ct = co.FFTree(log_n)
    n = 1 << log_n
    c = rand_fr_np(n, 11)
    e = ct.enter(c)
    assert (e == ct.eval(c)).all()
    assert (ct.exit(e) == c).all()
    low = c.copy()
    low[n // 2:] = 0  # degree < n/2: its values on D extend to its values on D'
    e2 = ct.enter(low)
    assert (ct.extend(np.ascontiguousarray(e2[0::2])) == e2[1::2]).all()
    assert (ct.extend(np.ascontiguousarray(e2[1::2]), 0, True) == e2[0::2]).all()
