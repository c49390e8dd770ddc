"""Host mirror of src/srs.rs: Trapdoor, SRS, verifier_runs_setup (in memory, and the cache_dir flavour that
reads the R1CS dump and writes the reference's SRS / precompute files) and verify.  Every vector stage runs
on the GPU through the C ABI; python ints only appear for the handful of trapdoor scalars."""
import os

from dataclasses import dataclass

import numpy as np

from . import curve, fr
from .gnark_r1cs import R1CSInstance, evaluate_monomial_basis_poly

P = fr.P


@dataclass
class Trapdoor:
    """src/srs.rs:41-50"""
    tau: int
    delta: int
    epsilon: int


@dataclass
class SRS:
    """src/srs.rs:30-39; bases as affine [n,8] uint64 + infinity masks."""
    g_m: tuple
    g_q: tuple
    g_k: tuple  # three (xy, inf) pairs

    def as_list(self):
        return [self.g_m, self.g_q, self.g_k[0], self.g_k[1], self.g_k[2]]


def srs_scalars(prover, inst: R1CSInstance, td: Trapdoor):
    """The scalars k with base = k*G for every SRS vector of an instance that exists only in python (bench.py's synthetic
    circuits, the tests): verifier_runs_setup + compute_srs_matrices, src/srs.rs:112-167,177-361 (Lagrange bases at tau through
    the barycentric formula, src/ec_fft.rs:340-390,424-450; accumulate_m_values, src/srs.rs:53-84).  ONE implementation since
    round 5: the device pipeline of dvp_setup_cache_dir (csrc/setup.hip: setup_scalars_core), fed this instance's CSR matrices
    through dvp_setup_scalars; rounds 1-4 orchestrated the same stages from here, vector by vector over the host-pointer seams
    (srs_scalars_hostside below keeps that route as the cross-check the tests compare against)."""
    import ctypes as C

    from ._native import lib, check, ptr

    assert td.tau % P and td.delta % P and td.epsilon % P  # src/srs.rs:199-201
    m = inst.num_constraints
    log_m = m.bit_length() - 1
    t, d, e = (fr.limbs(x) for x in (td.tau, td.delta, td.epsilon))
    mats = (inst.l, inst.r, inst.o)
    keep = [[np.ascontiguousarray(getattr(mt, f), dtype=np.uint32) for mt in mats] for f in ("row_ptr", "wire", "coeff")]
    arr = [(C.c_void_p * 3)(*[a.ctypes.data for a in col]) for col in keep]
    coeffs = np.ascontiguousarray(inst.coeffs, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((inst.n_wires + 5 * m, 4), dtype=np.uint64)
    check(lib.dvp_setup_scalars(ptr(t), ptr(d), ptr(e), log_m, inst.num_public_inputs, inst.n_rows, inst.n_wires, ptr(coeffs), coeffs.shape[0],
                                arr[0], arr[1], arr[2], ptr(out), out.shape[0]), "dvp_setup_scalars")
    o = inst.n_wires
    return out[:o], out[o:o + m], [out[o + m:o + 2 * m], out[o + 2 * m:o + 3 * m], out[o + 3 * m:]]


def srs_scalars_hostside(prover, inst: R1CSInstance, td: Trapdoor):
    """rounds 1-4's orchestration of the same stages from python (every vector through the host-pointer seams): kept as the
    independent cross-check of srs_scalars / dvp_setup_cache_dir in the tests"""
    assert td.tau % P and td.delta % P and td.epsilon % P  # src/srs.rs:199-201
    m = inst.num_constraints
    d, d2 = prover.domains()
    bar, z2inv = prover.domain_tables(0)      # 1/Z_D'(D_i), 1/Z_D(D'_i)
    bard, z2dinv = prover.domain_tables(1)    # 1/Z_D''(D'_i), 1/Z_D'(D_i)
    z_tau = prover.vanish_at(0, td.tau)
    zd_tau = prover.vanish_at(1, td.tau)
    assert z_tau and zd_tau
    delta2 = td.delta * td.delta % P
    # L_i(tau) = Z(tau) / ((tau - s_i) Z'(s_i))
    l_tau = fr.scale(fr.mul(fr.batch_inverse(fr.scalar_sub(td.tau, d)), bar), z_tau)
    l_taud = fr.scale(fr.mul(fr.batch_inverse(fr.scalar_sub(td.tau, d2)), bard), zd_tau)
    l_taul = np.empty((2 * m, 4), dtype=np.uint64)  # unified domain, interleaved (src/ec_fft.rs:445-448)
    l_taul[0::2] = fr.scale(fr.mul(l_tau, z2dinv), zd_tau)
    l_taul[1::2] = fr.scale(fr.mul(l_taud, z2inv), z_tau)
    # m_j(tau, delta) = sum_i (A_ij + delta B_ij + delta^2 C'_ij) L_i(tau), C' = C - D (Vandermonde on the public wires)
    lt_rows = l_tau[: inst.n_rows]
    mv = fr.spmv(*_t(inst.l, inst.n_wires), inst.coeffs, lt_rows)
    mb = fr.spmv(*_t(inst.r, inst.n_wires), inst.coeffs, lt_rows)
    mc = fr.spmv(*_t(inst.o, inst.n_wires), inst.coeffs, lt_rows)
    m_vals = fr.axpy(fr.axpy(mv, td.delta, mb), delta2, mc)  # whole vectors on the GPU: n_wires is 2^23 for the SP1 circuit
    pw = np.zeros((m, 4), dtype=np.uint64)  # d_i^0
    pw[:, 0] = 1
    for j in range(inst.num_public_inputs):  # -d_i^j on wire 1+j of every row (src/gnark_r1cs.rs:333-386)
        m_vals[1 + j] = fr.limbs((fr.to_int(m_vals[1 + j]) - delta2 * fr.dot(pw, l_tau)) % P)
        pw = fr.mul(pw, d)
    g_m = fr.scale(m_vals, td.epsilon)
    g_q = fr.scale(l_taud, z_tau * delta2 % P * td.epsilon % P)
    g_k = [l_tau, fr.scale(l_tau, td.delta), fr.scale(l_taul, delta2)]
    return g_m, g_q, g_k


def verifier_runs_setup_cache_dir(td: Trapdoor, cache_dir, num_public_inputs: int, write_precomputes: bool = True,
                                  return_scalars: bool = False):
    """SRS::verifier_runs_setup(trapdoor, cache_dir, num_public_inputs, ..), src/srs.rs:177-361, as the reference
    runs it: ONE library call (dvp_setup_cache_dir, csrc/setup.hip) reads cache_dir/r1cs_to_dvsnark, computes the SRS
    scalars on the device, writes g_m, g_q, g_k_0..2 as point-vector files (compute_srs_matrices ->
    write_point_vec_to_file) and, with write_precomputes, the domain files a reference prover/verifier would otherwise
    spend hours on (z_poly, z_polyd, bar_wts, bar_wtsd, z_vals2inv, z_vals2dinv).
    Returns (instance, prover context with the SRS loaded from the files just written)."""
    import ctypes as C

    from . import artifacts as A, io_utils
    from ._native import lib, check, ptr
    from .proving import Prover

    os.makedirs(cache_dir, exist_ok=True)
    t, d, e = (fr.limbs(x) for x in (td.tau, td.delta, td.epsilon))
    scalars = None
    if return_scalars:  # the discrete logs of the bases just written (tests pin the proof's commitments with them)
        nw, lg = C.c_uint32(0), C.c_uint32(0)
        inst = R1CSInstance.from_dump_file(os.path.join(cache_dir, A.R1CS_CONSTRAINTS_FILE), num_public_inputs)
        m = inst.num_constraints
        buf = np.zeros((inst.n_wires + 5 * m, 4), dtype=np.uint64)
        check(lib.dvp_setup_cache_dir_ex(ptr(t), ptr(d), ptr(e), os.fsencode(cache_dir), num_public_inputs, int(write_precomputes),
                                         ptr(buf), buf.shape[0], C.byref(nw), C.byref(lg)), "dvp_setup_cache_dir")
        assert nw.value == inst.n_wires and (1 << lg.value) == m
        o = inst.n_wires
        scalars = (buf[:o], buf[o:o + m], [buf[o + m:o + 2 * m], buf[o + 2 * m:o + 3 * m], buf[o + 3 * m:]])
    else:
        check(lib.dvp_setup_cache_dir(ptr(t), ptr(d), ptr(e), os.fsencode(cache_dir), num_public_inputs, int(write_precomputes)),
              "dvp_setup_cache_dir")
        inst = R1CSInstance.from_dump_file(os.path.join(cache_dir, A.R1CS_CONSTRAINTS_FILE), num_public_inputs)
    pv = Prover(inst)
    for which, name in enumerate(A.SRS_FILES):
        pv.set_srs_encoded(which, io_utils.read_point_vec_payload(os.path.join(cache_dir, name)))
    if return_scalars:
        return inst, pv, scalars
    return inst, pv


def _t(csr, n_wires):
    t = csr.transpose(n_wires)
    return t.row_ptr, t.wire, t.coeff


def verifier_runs_setup(prover, inst: R1CSInstance, td: Trapdoor) -> SRS:
    """SRS::verifier_runs_setup, src/srs.rs:177-361: the 6m fixed-base multiplications of
    compute_srs_matrices (:126-160) run as one batched GPU kernel per vector."""
    g_m, g_q, g_k = srs_scalars(prover, inst, td)
    mg = curve.point_scalar_mul_gen_batch
    return SRS(mg(g_m), mg(g_q), tuple(mg(v) for v in g_k))


def verify(td: Trapdoor, public_inputs, proof) -> bool:
    """SRS::verify, src/srs.rs:374-428."""
    from .proving import Proof, transcript_challenge

    pr = proof if isinstance(proof, Proof) else Proof.from_bytes(proof)
    try:
        pts, inf = curve.from_bytes(np.frombuffer(pr.commit_p + pr.kzg_k, dtype=np.uint8).reshape(2, 30))
    except Exception:
        return False
    a0, ok_a = pr.a0_fr()
    b0, ok_b = pr.b0_fr()
    if not (ok_a and ok_b):
        return False
    alpha = transcript_challenge(pr.commit_p, public_inputs)
    i0 = evaluate_monomial_basis_poly(public_inputs, alpha)
    r0 = (a0 * b0 - i0) % P
    delta2 = td.delta * td.delta % P
    u0 = (a0 + td.delta * b0 + delta2 * r0) % P * td.epsilon % P
    v0 = (td.tau - alpha) * td.epsilon % P
    gxy, ginf = curve.point_scalar_mul_gen_batch(fr.vec([1]))
    bases = np.stack([pts[1], gxy[0]])
    binf = np.array([inf[1], 0], dtype=np.uint8)
    lhs, lhs_inf = curve.multi_scalar_mul(fr.vec([v0, u0]), bases, binf)
    if lhs_inf or inf[0]:
        return bool(lhs_inf) and bool(inf[0])
    return bool((lhs == pts[0]).all())
