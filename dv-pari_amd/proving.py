"""Host mirror of src/proving.rs: Proof (to_bits/from_bits), Transcript challenge, and Proof::prove
driven through the GPU prover context (dvp_prover_* / dvp_prove in include/dvpari.h)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import fr
from ._native import lib, check, ptr
from .gnark_r1cs import R1CSInstance

P = fr.P


@dataclass
class Proof:
    """src/proving.rs:40-50"""
    commit_p: bytes  # 30
    kzg_k: bytes     # 30
    a0: bytes        # 29 = 232 LE bits (FrBits)
    b0: bytes

    def to_bytes(self) -> bytes:
        return self.commit_p + self.kzg_k + self.a0 + self.b0

    @staticmethod
    def from_bytes(b: bytes):
        assert len(b) == 118
        return Proof(b[:30], b[30:60], b[60:89], b[89:118])

    def to_bits(self):
        """Proof::to_bits, src/proving.rs:691-718 (LE bit order inside each byte)."""
        return [(byte >> i) & 1 for byte in self.to_bytes() for i in range(8)]

    @staticmethod
    def from_bits(bits):
        """Proof::from_bits, src/proving.rs:721-770."""
        assert len(bits) == 944
        by = bytes(sum(bits[8 * i + j] << j for j in range(8)) for i in range(118))
        return Proof.from_bytes(by)

    @staticmethod
    def prove(cache_dir, public_inputs, private_inputs) -> "Proof":
        """Proof::prove(cache_dir, public_inputs, private_inputs), src/proving.rs:426-688, with the reference's own
        signature: the R1CS dump and the SRS point files are read from cache_dir on the first call and stay in HBM
        (dvp_prove_cache_dir); release_cache_dir() frees them."""
        import os

        pub = _fr_arg(public_inputs)
        prv = _fr_arg(private_inputs)
        out = np.zeros(118, dtype=np.uint8)
        check(lib.dvp_prove_cache_dir(os.fspath(cache_dir).encode(), ptr(pub), pub.shape[0], ptr(prv), prv.shape[0], ptr(out)),
              "dvp_prove_cache_dir")
        return Proof.from_bytes(out.tobytes())

    def a0_fr(self):
        v = int.from_bytes(self.a0, "little")
        return (v, True) if v < P else (0, False)  # FrBits::to_fr, src/curve.rs:43-59

    def b0_fr(self):
        v = int.from_bytes(self.b0, "little")
        return (v, True) if v < P else (0, False)


def _fr_arg(v) -> np.ndarray:
    a = v if isinstance(v, np.ndarray) else (fr.vec(v) if len(v) else np.zeros((0, 4), dtype=np.uint64))
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


def release_cache_dir(cache_dir=None):
    """drop the prover(s) dvp_prove_cache_dir keeps for cache_dir (None: all)"""
    import os

    lib.dvp_cache_dir_release(None if cache_dir is None else os.fspath(cache_dir).encode())


PREP_WROTE_TREE2N, PREP_WROTE_BAR_WTS, PREP_WROTE_Z_VALS2INV, PREP_Z_POLY_NOT_MONIC = 0x1, 0x2, 0x4, 0x8
PREP_BAD_Z_POLY, PREP_BAD_BAR_WTS, PREP_BAD_Z_VALS2INV, PREP_BAD_TREE2N = 0x100, 0x200, 0x400, 0x800


def prover_prepares_precomputes(cache_dir, validate_precompute: bool = False) -> int:
    """prover_prepares_precomputes, src/proving.rs:225-325, through the C entry of the same name: needs z_poly in
    cache_dir (its length fixes m), reads tree2n or writes a minimal one, produces bar_wts and z_vals2inv when they are
    missing -- from the isogeny chain in milliseconds.  The GPU prover itself regenerates all of these on the device and
    reads none of the files; this exists so that a cache_dir prepared here is complete for a reference prover as well.
    validate_precompute: z_poly must vanish on D (the reference's asserts, :281-296) and every file found is compared
    with the regenerated values.  Returns the PREP_* report bits; a failed validation raises ValueError naming the file."""
    import os

    rep = C.c_uint32(0)
    rc = lib.dvp_prover_prepares_precomputes(os.fspath(cache_dir).encode(), int(bool(validate_precompute)), C.byref(rep))
    bad = [n for n, b in (("z_poly", PREP_BAD_Z_POLY), ("bar_wts", PREP_BAD_BAR_WTS), ("z_vals2inv", PREP_BAD_Z_VALS2INV),
                          ("tree2n", PREP_BAD_TREE2N)) if rep.value & b]
    if bad:
        msg = "vanishing poly does not evaluate to zero at all points in domain" if bad == ["z_poly"] else "differs from the regenerated values"
        raise ValueError(f"{cache_dir}: {', '.join(bad)}: {msg} (first bad index {lib.dvp_last_error_index()})")
    check(rc, "dvp_prover_prepares_precomputes")
    return rep.value


def transcript_challenge(commit_p: bytes, public_inputs) -> int:
    """Transcript::output, src/proving.rs:164-197."""
    cp = np.frombuffer(commit_p, dtype=np.uint8).copy()
    pub = fr.vec(public_inputs) if len(public_inputs) else np.zeros((1, 4), dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    check(lib.dvp_transcript_challenge(ptr(cp), ptr(pub), len(public_inputs), ptr(out)), "dvp_transcript_challenge")
    return fr.to_int(out)


def blake3(data: bytes) -> bytes:
    d = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, dtype=np.uint8)
    out = np.zeros(32, dtype=np.uint8)
    check(lib.dvp_blake3(ptr(d), len(data), ptr(out)), "dvp_blake3")
    return out.tobytes()


class Prover:
    """The artefacts Proof::prove reads from cache_dir (R1CS dump, SRS point vectors, TREE_2N and the
    prover precomputes, src/proving.rs:435-511,562-565,666-672), loaded once and kept in HBM."""

    def __init__(self, inst: R1CSInstance):
        self.inst = inst
        self.m = inst.num_constraints
        self.log_m = self.m.bit_length() - 1
        h = C.c_void_p()
        check(lib.dvp_prover_create(self.log_m, inst.num_public_inputs, inst.n_wires, C.byref(h)), "dvp_prover_create")
        self._h = h
        check(lib.dvp_prover_set_coeffs(h, ptr(inst.coeffs), inst.coeffs.shape[0]), "dvp_prover_set_coeffs")
        for which, mat in enumerate((inst.l, inst.r, inst.o)):
            check(lib.dvp_prover_set_matrix(h, which, inst.n_rows, ptr(mat.row_ptr), ptr(mat.wire), ptr(mat.coeff)),
                  "dvp_prover_set_matrix")

    @staticmethod
    def of_cache_dir(cache_dir, num_public_inputs: int, m: int) -> "Prover":
        """the prover Proof.prove(cache_dir, ..) keeps for cache_dir (opened if need be): a borrowed handle for
        inspection (debug reads, MSM plans), released by release_cache_dir, never closed from here"""
        import os

        self = Prover.__new__(Prover)
        self.inst, self.m, self.log_m, self._borrowed = None, m, m.bit_length() - 1, True
        h = C.c_void_p()
        check(lib.dvp_cache_dir_prover(os.fspath(cache_dir).encode(), num_public_inputs, C.byref(h)), "dvp_cache_dir_prover")
        self._h = h
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                lib.dvp_prover_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- SRS -------------------------------------------------------------------------------------
    def set_srs(self, srs):
        for which, (xy, inf) in enumerate(srs.as_list()):
            xy = np.ascontiguousarray(xy, dtype=np.uint64)
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
            check(lib.dvp_prover_set_srs_affine(self._h, which, ptr(xy), ptr(inf), xy.shape[0]), "dvp_prover_set_srs_affine")

    def set_srs_encoded(self, which: int, enc: np.ndarray):
        """payload of a reference point-vector file (n x 30 bytes, src/io_utils.rs:83-111)"""
        e = np.ascontiguousarray(enc, dtype=np.uint8).reshape(-1, 30)
        check(lib.dvp_prover_set_srs_encoded(self._h, which, ptr(e), e.shape[0]), "dvp_prover_set_srs_encoded")

    # ---- domain data -------------------------------------------------------------------------------
    def domains(self):
        d = np.zeros((self.m, 4), dtype=np.uint64)
        d2 = np.zeros((self.m, 4), dtype=np.uint64)
        check(lib.dvp_prover_domains(self._h, ptr(d), ptr(d2)), "dvp_prover_domains")
        return d, d2

    def domain_tables(self, which: int):
        a = np.zeros((self.m, 4), dtype=np.uint64)
        b = np.zeros((self.m, 4), dtype=np.uint64)
        check(lib.dvp_prover_domain_tables(self._h, which, ptr(a), ptr(b)), "dvp_prover_domain_tables")
        return a, b

    def vanish_at(self, which: int, x: int) -> int:
        """Z_D(x) / Z_D'(x) through the isogeny chain"""
        from .ec_fft import FFTree  # noqa: F401  (same chain as dvp_ecfft_vanish_at, on the prover's own tree)
        if not hasattr(self, "_tree"):
            self._tree = FFTree(2 * self.m)
        out = np.zeros(4, dtype=np.uint64)
        check(lib.dvp_ecfft_vanish_at(self._tree._h, which, ptr(fr.limbs(x)), ptr(out)), "dvp_ecfft_vanish_at")
        return fr.to_int(out)

    # ---- prove -------------------------------------------------------------------------------------
    def prove(self, public_inputs, private_inputs) -> Proof:
        """Proof::prove(cache_dir, public_inputs, private_inputs), src/proving.rs:426-688."""
        pub, prv = _fr_arg(public_inputs), _fr_arg(private_inputs)
        out = np.zeros(118, dtype=np.uint8)
        check(lib.dvp_prove(self._h, ptr(pub), pub.shape[0], ptr(prv), prv.shape[0], ptr(out)), "dvp_prove")
        return Proof.from_bytes(out.tobytes())

    def debug(self, name: str, n: int = None):
        if n is None:
            n = 1 if name in ("alpha", "a0", "b0", "i0", "r0") else (2 * self.m if name == "kr" else self.m)
        out = np.zeros((n, 4), dtype=np.uint64)
        check(lib.dvp_prover_debug_read(self._h, name.encode(), ptr(out), n), "dvp_prover_debug_read")
        return out

    def transcript_dev(self, commit_p: bytes, public_inputs):
        """(alpha, -Z_D(alpha)) from the DEVICE flavour of the transcript (k_transcript / k_zalpha): parity-test access"""
        pub = _fr_arg(public_inputs)
        cp = np.frombuffer(bytes(commit_p), dtype=np.uint8).copy()
        a, z = np.zeros((1, 4), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64)
        check(lib.dvp_prover_debug_transcript_dev(self._h, ptr(cp), ptr(pub) if pub.shape[0] else None, pub.shape[0], ptr(a), ptr(z)),
              "dvp_prover_debug_transcript_dev")
        return fr.to_int(a), fr.to_int(z)

    # ---- device-resident / phased flavours --------------------------------------------------------------
    def prove_dev(self, d_assignment: int, stream: int = 0) -> Proof:
        """assignment = [1, public.., private..] as n_wires x 4 uint64 already in HBM (device pointer)."""
        out = np.zeros(118, dtype=np.uint8)
        check(lib.dvp_prove_dev(self._h, d_assignment, ptr(out), stream), "dvp_prove_dev")
        return Proof.from_bytes(out.tobytes())

    def begin(self, d_assignment: int, stream: int = 0, need_extend: bool = True):
        check(lib.dvp_prove_begin_partial(self._h, d_assignment, int(need_extend), stream), "dvp_prove_begin")

    def extend_count(self) -> int:
        """3 (a, b, c'; i by Horner) or 4: the vectors Proof::extend_evals extends (src/proving.rs:410-422)"""
        return int(lib.dvp_prover_extend_count(self._h))

    def extend_vectors(self, mask: int, stream: int = 0):
        check(lib.dvp_prove_extend_vectors(self._h, mask, stream), "dvp_prove_extend_vectors")

    def extended_ptr(self, v: int) -> int:
        out = C.c_void_p()
        check(lib.dvp_prover_extended_ptr(self._h, v, C.byref(out)), "dvp_prover_extended_ptr")
        return out.value

    def mark_extended(self, v: int):
        check(lib.dvp_prove_mark_extended(self._h, v), "dvp_prove_mark_extended")

    def quotient(self, stream: int = 0):
        check(lib.dvp_prove_quotient(self._h, stream), "dvp_prove_quotient")

    def msm_size(self, which: int) -> int:
        return int(lib.dvp_prover_msm_size(self._h, which))

    def msm_plan(self, which: int):
        """(window bits, windows) of the fixed-base MSM `which`, (0, 0) before its first use"""
        c, w = C.c_int(0), C.c_int(0)
        check(lib.dvp_prover_msm_plan(self._h, which, C.byref(c), C.byref(w)), "dvp_prover_msm_plan")
        return c.value, w.value

    def msm_table(self, which: int):
        """(bytes of HBM, signed binary windows -- the default flavour -- ?) of the fixed-base tables of MSM `which`"""
        s = C.c_int(0)
        b = int(lib.dvp_prover_msm_table_bytes(self._h, which, C.byref(s)))
        return b, bool(s.value)

    def msm_partial(self, which: int, lo: int, hi: int, d_out_xy: int, d_out_inf: int, stream: int = 0):
        check(lib.dvp_prover_msm_partial(self._h, which, lo, hi, d_out_xy, d_out_inf, stream), "dvp_prover_msm_partial")

    def challenge(self, d_commit_xy: int, d_commit_inf: int, stream: int = 0):
        check(lib.dvp_prove_challenge(self._h, d_commit_xy, d_commit_inf, stream), "dvp_prove_challenge")

    def challenge_partial(self, d_commit_xy: int, d_commit_inf: int, d_range, k_range, d_record: int, stream: int = 0):
        """phase 2, part 1 of an index-sharded prove: this rank's slice of the barycentric sums (d_range of D) and the inverse
        denominators its K-scalar range needs -> a 128-byte record at d_record (dvp_prove_challenge_partial)"""
        check(lib.dvp_prove_challenge_partial(self._h, d_commit_xy, d_commit_inf, d_range[0], d_range[1], k_range[0], k_range[1], d_record, stream),
              "dvp_prove_challenge_partial")

    def challenge_finish(self, d_records: int, n_records: int, k_range, stream: int = 0):
        """part 2: a0 b0 i0 r0 from the records of all ranks, then the K scalars of k_range only"""
        check(lib.dvp_prove_challenge_finish(self._h, d_records, n_records, k_range[0], k_range[1], stream), "dvp_prove_challenge_finish")

    def finish(self, d_kzg_xy: int, d_kzg_inf: int, stream: int = 0) -> Proof:
        out = np.zeros(118, dtype=np.uint8)
        check(lib.dvp_prove_finish(self._h, d_kzg_xy, d_kzg_inf, ptr(out), stream), "dvp_prove_finish")
        return Proof.from_bytes(out.tobytes())
