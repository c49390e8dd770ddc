"""Oracle-backed checks that stay affordable at full size (2^20 .. 2^22 constraints), used by the -m gpu tests.

Nothing here calls the product: every expected value comes from python big-int arithmetic (oracle/pyref.py) or the
OpenSSL-pinned C oracle, from vectors the prover exposes through dvp_prover_debug_read.

  * discrete-log identity (the shape of the reference's own test_msm, src/curve.rs:218-232): an MSM over bases
    k_i*G must equal (sum s_i k_i)*G, so commit_p / kzg_k are pinned by one oracle mulgen each
    (src/proving.rs:463,512,515,680);
  * the SRS scalars k_i (compute_srs_matrices, src/srs.rs:112-167) are themselves pinned by brute-force Lagrange
    products on sampled indices, and the domain by direct scalar multiplication on the ECFFT curve (src/ec_fft.rs:158-162);
  * a0, b0 by the barycentric formula in big ints (src/ec_fft.rs:455-491) and the designated-verifier equation on
    discrete logs (src/srs.rs:374-428).
"""
import random

import numpy as np

import pyref as o
import c_oracle as co
from util import from_limbs, np_dot_mod_fast

P = o.P


def oracle_leaf(log_leaves: int, i: int):
    """x(C + i*g), g = 2^(28-log_leaves)*GEN: leaf i of the 2^log_leaves-leaf tree (src/ec_fft.rs:116-119,158-162)"""
    g = o.ECFFT_GEN
    for _ in range(o.ECFFT_LOG_ORDER - log_leaves):
        g = o.sw_add(g, g, o.ECFFT_A)
    return o.sw_add(o.ECFFT_COSET, o.sw_mul(i, g, o.ECFFT_A), o.ECFFT_A)[0]


def prod_diff(x, dom, skip=None):
    """prod_j (x - dom[j]), j != skip"""
    acc = 1
    for j, d in enumerate(dom):
        if j != skip:
            acc = acc * (x - d) % P
    return acc


def lagrange_at(dom, i, tau):
    """L_i(tau) on `dom` by its definition (two products over the whole domain)"""
    return prod_diff(tau, dom, i) * o.fr_inv(prod_diff(dom[i], dom, i)) % P


def check_domains(pv, log_m, rnd, samples=6):
    d, d2 = pv.domains()
    D, D2 = from_limbs(d), from_limbs(d2)
    m = 1 << log_m
    for _ in range(samples):
        i = rnd.randrange(m)
        assert D[i] == oracle_leaf(log_m + 1, 2 * i), i       # D = even leaves, D' = odd leaves (src/ec_fft.rs:179-189)
        assert D2[i] == oracle_leaf(log_m + 1, 2 * i + 1), i
    return D, D2


def check_srs_scalars(inst, trap, D, D2, g_m, g_q, g_k, rnd, samples=2):
    """sampled entries of every SRS scalar vector against the definitions (src/srs.rs:112-167; C' = C - D with
    D_ij = d_i^j on the public wires, src/gnark_r1cs.rs:333-386)"""
    tau, delta, eps = trap
    m = len(D)
    delta2 = delta * delta % P
    z_tau = prod_diff(tau, D)
    U = [x for pair in zip(D, D2) for x in pair]  # unified domain, interleaved (src/ec_fft.rs:445-448)
    gk0, gk1, gk2, gq = (from_limbs(v) for v in (g_k[0], g_k[1], g_k[2], g_q))
    for _ in range(samples):
        i = rnd.randrange(m)
        li = lagrange_at(D, i, tau)
        assert gk0[i] == li, i
        assert gk1[i] == li * delta % P, i
        assert gq[i] == z_tau * delta2 % P * lagrange_at(D2, i, tau) % P * eps % P, i
        u = rnd.randrange(2 * m)
        assert gk2[u] == lagrange_at(U, u, tau) * delta2 % P, u
    # g_m on sampled private wires: eps * sum_i (A_ij + delta B_ij + delta^2 C_ij) L_i(tau), L_i(tau) = g_k_0[i]
    coeffs = from_limbs(inst.coeffs)
    gm = from_limbs(g_m)
    first_private = 1 + inst.num_public_inputs
    for _ in range(4 * samples):
        j = rnd.randrange(first_private, inst.n_wires)
        acc = 0
        for mat, scale in ((inst.l, 1), (inst.r, delta), (inst.o, delta2)):
            nnz = int(mat.row_ptr[inst.n_rows])
            hits = np.nonzero(mat.wire[:nnz] == j)[0]
            rows = np.searchsorted(mat.row_ptr, hits, side="right") - 1
            for h, r in zip(hits, rows):
                acc += scale * coeffs[int(mat.coeff[h])] % P * gk0[int(r)]
        assert gm[j] == acc % P * eps % P, j
    return z_tau


def check_proof(dvp, pv, inst, trap, pub, prv, proof, scalars, D=None, check_bary=True, rnd=None, w=None):
    """commit_p, kzg_k, alpha, a0, b0 of `proof` against the oracle, from the prover's scalar vectors and the SRS
    scalars (g_m, g_q, g_k) the bases were generated from; the assignment either as pub / prv int lists or as the
    limb array w = [1, pub.., prv..]"""
    g_m, g_q, g_k = scalars
    m = inst.num_constraints
    if w is None:
        w = dvp.fr.vec([1] + list(pub) + list(prv))
    assert w.shape[0] == inst.n_wires
    q2, ka, kb, kr = (pv.debug(k) for k in ("q2", "ka", "kb", "kr"))
    dl_commit = (np_dot_mod_fast(w, g_m) + np_dot_mod_fast(q2, g_q)) % P
    dl_kzg = (np_dot_mod_fast(ka, g_k[0]) + np_dot_mod_fast(kb, g_k[1]) + np_dot_mod_fast(kr, g_k[2])) % P
    assert proof.commit_p == co.xsk233_encode(co.k233_mulgen(dl_commit))
    assert proof.kzg_k == co.xsk233_encode(co.k233_mulgen(dl_kzg))
    alpha = o.transcript_challenge(proof.commit_p, pub)
    assert from_limbs(pv.debug("alpha"))[0] == alpha
    a0, ok_a = proof.a0_fr()
    b0, ok_b = proof.b0_fr()
    assert ok_a and ok_b
    if check_bary:
        # bar_wts[i] = 1 / Z'(d_i) (sampled by brute force), then P(alpha) = Z(alpha) sum y_i w_i / (alpha - d_i)
        bar = from_limbs(pv.debug("bar_wts"))
        for _ in range(2):
            i = rnd.randrange(m)
            assert bar[i] * prod_diff(D[i], D, i) % P == 1, i
        z_alpha = prod_diff(alpha, D)
        assert a0 == o.barycentric_eval(D, bar, z_alpha, from_limbs(pv.debug("a")), alpha)
        assert b0 == o.barycentric_eval(D, bar, z_alpha, from_limbs(pv.debug("b")), alpha)
    assert o.verify_dl(trap, pub, dl_commit, dl_kzg, a0, b0, alpha)
    return dl_commit, dl_kzg
