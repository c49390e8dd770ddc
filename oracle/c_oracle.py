"""ctypes loader for the C oracle (oracle/dvp_oracle.c).  ORACLE = test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libdvp_oracle.so")
OSSL_PATH = os.path.join(_HERE, "_build", "libdvp_oracle_ossl.so")
ECFFT_PATH = os.path.join(_HERE, "_build", "libdvp_oracle_ecfft.so")


def build(force=False):
    src = os.path.join(_HERE, "dvp_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libdvp_oracle.so"], stdout=subprocess.DEVNULL)
    src3 = os.path.join(_HERE, "dvp_oracle_ecfft.c")
    if force or not os.path.exists(ECFFT_PATH) or os.path.getmtime(ECFFT_PATH) < os.path.getmtime(src3):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libdvp_oracle_ecfft.so"], stdout=subprocess.DEVNULL)
    src2 = os.path.join(_HERE, "dvp_oracle_ossl.c")
    if force or not os.path.exists(OSSL_PATH) or os.path.getmtime(OSSL_PATH) < os.path.getmtime(src2):
        # the OpenSSL datapoint is optional: a box without libcrypto headers still gets the main oracle
        subprocess.call(["make", "-C", _HERE, "-B", "_build/libdvp_oracle_ossl.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB_PATH


def host_threads(cap: int = 64) -> int:
    """threads this process may really use: the affinity mask clipped by the cgroup CPU quota (a one-GPU box shows the
    whole host in its mask but owns 16 CPUs of it)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        _lib.dvo_msm.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp, C.POINTER(C.c_int)]
        _lib.dvo_msm.restype = C.c_int
        _lib.dvo_msm_pippenger.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp, C.POINTER(C.c_int)]
        _lib.dvo_msm_pippenger.restype = C.c_int
        _lib.dvo_k233_mul.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int)]
        _lib.dvo_k233_mulgen.argtypes = [vp, vp, C.POINTER(C.c_int)]
        _lib.dvo_k233_add.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.POINTER(C.c_int)]
        _lib.dvo_tau_digits.argtypes = [vp, vp]
        _lib.dvo_tau_digits.restype = C.c_int
        _lib.dvo_tau_digits_fast.argtypes = [vp, vp]
        _lib.dvo_tau_digits_fast.restype = C.c_int
        _lib.dvo_xsk233_encode.argtypes = [vp, C.c_int, vp]
        _lib.dvo_xsk233_decode.argtypes = [vp, vp, C.POINTER(C.c_int)]
        _lib.dvo_xsk233_decode.restype = C.c_int
        _lib.dvo_fr_mont_mul.argtypes = [vp, vp, vp]
        _lib.dvo_fr_butterfly_passes.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
        _lib.dvo_fr_butterfly_passes.restype = C.c_int
        _lib.dvo_fr_pointwise_stages.argtypes = [vp, C.c_size_t, C.c_int]
        _lib.dvo_fr_pointwise_stages.restype = C.c_int
        _lib.dvo_fr_convert.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
        _lib.dvo_fr_extend.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp]
        _lib.dvo_fr_extend.restype = C.c_int
        _lib.dvo_prove_load.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        _lib.dvo_prove_load.restype = C.c_int
        _lib.dvo_prove_free.argtypes = []
        _lib.dvo_prove_commit.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t, vp, vp, vp, C.POINTER(C.c_int), vp]
        _lib.dvo_prove_commit.restype = C.c_long
        _lib.dvo_prove_open.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_int), vp]
        _lib.dvo_prove_open.restype = C.c_int
        for f in ("dvo_gf_mul",):
            getattr(_lib, f).argtypes = [vp, vp, vp]
        for f in ("dvo_gf_sqr", "dvo_gf_inv"):
            getattr(_lib, f).argtypes = [vp, vp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _limbs(v, n=4):
    return np.frombuffer(int(v).to_bytes(8 * n, "little"), dtype="<u8").copy()


def _int(a):
    return int.from_bytes(np.ascontiguousarray(a, dtype="<u8").tobytes(), "little")


def gf_mul(a, b):
    o = np.zeros(4, dtype=np.uint64)
    lib().dvo_gf_mul(_p(_limbs(a)), _p(_limbs(b)), _p(o))
    return _int(o)


def gf_sqr(a):
    o = np.zeros(4, dtype=np.uint64)
    lib().dvo_gf_sqr(_p(_limbs(a)), _p(o))
    return _int(o)


def gf_inv(a):
    o = np.zeros(4, dtype=np.uint64)
    lib().dvo_gf_inv(_p(_limbs(a)), _p(o))
    return _int(o)


def _pt_out(o, inf):
    return None if inf.value else (_int(o[:4]), _int(o[4:]))


def _pt_in(pt):
    if pt is None:
        return np.zeros(8, dtype=np.uint64), 1
    return np.concatenate([_limbs(pt[0]), _limbs(pt[1])]), 0


def k233_mul(k, pt, frob=True, tnaf5=False):
    """k * pt by one of three independent routes: integer double-and-add (frob=False), tau-adic 4-digit windows (default),
    width-5 tau-NAF (tnaf5=True: what the timed reference-shaped MSM uses)"""
    a, inf = _pt_in(pt)
    o = np.zeros(8, dtype=np.uint64)
    oi = C.c_int(0)
    lib().dvo_k233_mul(_p(_limbs(k)), _p(a), inf, 2 if tnaf5 else (1 if frob else 0), _p(o), C.byref(oi))
    return _pt_out(o, oi)


def k233_mulgen(k):
    o = np.zeros(8, dtype=np.uint64)
    oi = C.c_int(0)
    lib().dvo_k233_mulgen(_p(_limbs(k)), _p(o), C.byref(oi))
    return _pt_out(o, oi)


def tau_digits(k):
    d = np.zeros(260, dtype=np.uint8)
    n = lib().dvo_tau_digits(_p(_limbs(k)), _p(d))
    return [int(x) for x in d[:n]]


def tau_digits_fast(k):
    """the expansion the timed CPU baselines use (fixed-point quotients, as the GPU kernel rounds)"""
    d = np.zeros(260, dtype=np.uint8)
    n = lib().dvo_tau_digits_fast(_p(_limbs(k)), _p(d))
    return [int(x) for x in d[:n]]


def msm(scalars: np.ndarray, bases: np.ndarray, inf: np.ndarray = None, threads: int = 1):
    """scalars [n,4] u64, bases [n,8] u64 -> (point or None).  Reference shape: src/curve.rs:141-158."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    b = np.ascontiguousarray(bases, dtype=np.uint64)
    ip = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        ip = _p(inf)
    o = np.zeros(8, dtype=np.uint64)
    oi = C.c_int(0)
    lib().dvo_msm(_p(s), _p(b), ip, s.shape[0], threads, _p(o), C.byref(oi))
    return _pt_out(o, oi)


def msm_pippenger(scalars: np.ndarray, bases: np.ndarray, inf: np.ndarray = None, threads: int = 1):
    """the same sum by a host-side bucket method ("best CPU"): tau-adic windows, per-thread bucket sets"""
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    b = np.ascontiguousarray(bases, dtype=np.uint64)
    ip = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        ip = _p(inf)
    o = np.zeros(8, dtype=np.uint64)
    oi = C.c_int(0)
    lib().dvo_msm_pippenger(_p(s), _p(b), ip, s.shape[0], threads, _p(o), C.byref(oi))
    return _pt_out(o, oi)


def xsk233_encode(pt) -> bytes:
    a, inf = _pt_in(pt)
    o = np.zeros(30, dtype=np.uint8)
    lib().dvo_xsk233_encode(_p(a), inf, _p(o))
    return o.tobytes()


def xsk233_decode(buf: bytes):
    i = np.frombuffer(buf, dtype=np.uint8).copy()
    o = np.zeros(8, dtype=np.uint64)
    oi = C.c_int(0)
    ok = lib().dvo_xsk233_decode(_p(i), _p(o), C.byref(oi))
    return (_pt_out(o, oi) if ok else None), bool(ok)


def fr_mont_mul(a, b):
    """a * b / 2^256 mod p (the Montgomery product the extend butterflies are made of)"""
    o = np.zeros(4, dtype=np.uint64)
    lib().dvo_fr_mont_mul(_p(_limbs(a)), _p(_limbs(b)), _p(o))
    return _int(o)


def fr_butterfly_passes(data: np.ndarray, mats: np.ndarray, passes: int, threads: int = 1):
    """in place: `passes` extend-shaped butterfly passes over data [n,4] with matrices mats [2, n/2, 4, 4] (Montgomery)"""
    assert data.flags["C_CONTIGUOUS"] and mats.flags["C_CONTIGUOUS"] and mats.shape[:2] == (2, data.shape[0] // 2)
    return lib().dvo_fr_butterfly_passes(_p(data), _p(mats), data.shape[0], passes, threads)


def fr_pointwise_stages(m: int, threads: int = 1, seed: int = 3):
    """runs the pointwise Fr stages of Proof::prove (src/proving.rs:492-654: quotient, three barycentric evaluations with
    their own batch inversions, denominators, K scalars) on synthetic m-length vectors; returns the wall time in seconds"""
    import time

    rng = np.random.default_rng(seed)
    buf = rng.integers(1, 2**62, size=(21 * m, 4), dtype=np.uint64)
    buf[:, 3] &= np.uint64((1 << 38) - 1)
    t0 = time.perf_counter()
    lib().dvo_fr_pointwise_stages(_p(buf), m, threads)
    return time.perf_counter() - t0


_ossl = None


def fr_extend(evals: np.ndarray, dec: np.ndarray, rec: np.ndarray, threads: int = 1) -> np.ndarray:
    """FFTree::extend(evals, Moiety::S1) over explicit butterfly matrices ((n - 1) x 4 canonical Fr each, layer d at matrix offset
    n - (n >> d)): the loop dvo_prove_commit runs, exposed so that it can be pinned against the recursive oracle extend"""
    e = np.ascontiguousarray(evals, dtype=np.uint64)
    n = e.shape[0]
    d, r = np.ascontiguousarray(dec, dtype=np.uint64), np.ascontiguousarray(rec, dtype=np.uint64)
    assert d.shape == r.shape == ((n - 1) * 4, 4), (d.shape, n)
    out = np.zeros_like(e)
    assert lib().dvo_fr_extend(_p(e), _p(d), _p(r), n, threads, _p(out)) == 0
    return out


class ProveInputs:
    """what Proof::prove reads from its cache_dir, as arrays (canonical Fr = [n, 4] uint64; points affine [n, 8] uint64)"""

    def __init__(self, m, n_wires, n_pub, csr, coeffs, n_rows, d, d2, bar_wts, z_vals2inv, z_poly, dec, rec, bases_a, bases_k):
        self.m, self.n_wires, self.n_pub, self.n_rows = m, n_wires, n_pub, n_rows
        self.csr = [tuple(np.ascontiguousarray(x, dtype=np.uint32) for x in t) for t in csr]  # 3 x (row_ptr, wire, coeff id)
        self.coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        self.tabs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (d, d2, bar_wts, z_vals2inv, z_poly, dec, rec)]
        self.bases_a = np.ascontiguousarray(bases_a, dtype=np.uint64)
        self.bases_k = np.ascontiguousarray(bases_k, dtype=np.uint64)
        assert self.bases_a.shape == (n_wires + m, 8) and self.bases_k.shape == (4 * m, 8)
        assert self.tabs[4].shape == (m + 1, 4) and self.tabs[5].shape == ((m - 1) * 4, 4)


def prove_cpu(inp: ProveInputs, public_inputs, private_inputs, transcript, threads: int = 1):
    """Proof::prove (src/proving.rs:426-688) end to end on the CPU: dvo_prove_commit -> transcript(commit_p bytes) -> dvo_prove_open.
    Returns (commit_p bytes, kzg_k bytes, a0, b0, {stage: seconds}); raises ValueError(row) on an unsatisfied constraint."""
    L = lib()
    t = inp.tabs
    assert L.dvo_prove_load(inp.m, inp.n_wires, inp.n_pub, _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]), _p(t[4]), _p(t[5]), _p(t[6]), threads) == 0
    try:
        w = np.zeros((inp.n_wires, 4), dtype=np.uint64)
        vals = [1] + [int(x) for x in public_inputs] + [int(x) for x in private_inputs]
        assert len(vals) == inp.n_wires
        w[:] = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype="<u8").reshape(-1, 4)
        arr3 = lambda k: (C.c_void_p * 3)(*[inp.csr[j][k].ctypes.data for j in range(3)])
        rp, wi, ci = arr3(0), arr3(1), arr3(2)
        xy = np.zeros(8, dtype=np.uint64)
        inf = C.c_int(0)
        st1 = np.zeros(4, dtype=np.float64)
        rc = L.dvo_prove_commit(rp, wi, ci, _p(inp.coeffs), inp.coeffs.shape[0], inp.n_rows, _p(w), _p(inp.bases_a), _p(xy), C.byref(inf), _p(st1))
        if rc < 0:
            raise ValueError(-1 - rc)
        commit = xsk233_encode(_pt_out(xy, inf))
        alpha = int(transcript(commit))
        a0, b0, kxy = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        kinf = C.c_int(0)
        st2 = np.zeros(3, dtype=np.float64)
        assert L.dvo_prove_open(_p(_limbs(alpha)), _p(inp.bases_k), _p(a0), _p(b0), _p(kxy), C.byref(kinf), _p(st2)) == 0
        kzg = xsk233_encode(_pt_out(kxy, kinf))
        stages = {"matvec_sequential": st1[0], "extend_x4": st1[1], "quotient": st1[2], "msm_commit": st1[3],
                  "barycentric_x3_sequential": st2[0], "inversions_kscalars": st2[1], "msm_k": st2[2]}
        return commit, kzg, _int(a0), _int(b0), {k: float(v) for k, v in stages.items()}
    finally:
        L.dvo_prove_free()


def openssl_msm(scalars: np.ndarray, bases: np.ndarray, threads: int = 1):
    """the reference's MSM shape on OpenSSL's sect233k1 (EC_POINT_mul per point + add); None if libcrypto is unavailable"""
    global _ossl
    if _ossl is None:
        build()
        if not os.path.exists(OSSL_PATH):
            return NotImplemented
        _ossl = C.CDLL(OSSL_PATH)
        _ossl.dvo_openssl_msm.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        _ossl.dvo_openssl_msm.restype = C.c_int
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    b = np.ascontiguousarray(bases, dtype=np.uint64)
    o = np.zeros(8, dtype=np.uint64)
    oi = C.c_int(0)
    if _ossl.dvo_openssl_msm(_p(s), _p(b), s.shape[0], threads, _p(o), C.byref(oi)) != 0:
        raise RuntimeError("dvo_openssl_msm failed")
    return _pt_out(o, oi)


# ---- ECFFT restatement in C (oracle/dvp_oracle_ecfft.c): pyref.FFTree at sizes python cannot reach -------------------
_ecfft = None


def _ecfft_lib():
    global _ecfft
    if _ecfft is None:
        build()
        _ecfft = C.CDLL(ECFFT_PATH)
        vp = C.c_void_p
        _ecfft.dvo_ecfft_set_threads.argtypes = [C.c_int]
        _ecfft.dvo_ecfft_set_threads(host_threads())
        _ecfft.dvo_fftree_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp]
        _ecfft.dvo_fftree_new.restype = vp
        _ecfft.dvo_fftree_free.argtypes = [vp]
        _ecfft.dvo_fftree_layer.argtypes = [vp, C.c_int, vp]
        for f in ("dvo_fftree_enter", "dvo_fftree_exit", "dvo_fftree_eval"):
            getattr(_ecfft, f).argtypes = [vp, vp, C.c_int, vp]
            getattr(_ecfft, f).restype = C.c_int
        _ecfft.dvo_fftree_extend.argtypes = [vp, vp, C.c_int, C.c_int, vp]
        _ecfft.dvo_fftree_extend.restype = C.c_int
        _ecfft.dvo_fftree_matrices.argtypes = [vp, C.c_int, C.c_int, vp]
        _ecfft.dvo_fftree_matrices.restype = C.c_int
        _ecfft.dvo_fftree_vanish_even_at.argtypes = [vp, vp, C.c_int, vp]
    return _ecfft


class FFTree:
    """pyref.FFTree in C: arrays are numpy uint64 [n,4] canonical limbs; `sl` = log2 of pyref's `stride`."""

    def __init__(self, log_n: int, shifted: bool = False, base_log_n: int = None):
        import pyref as o

        self.log_n, self.n = log_n, 1 << log_n
        consts = np.concatenate([_limbs(v) for v in (o.ECFFT_A, o.ECFFT_GEN[0], o.ECFFT_GEN[1], o.ECFFT_COSET[0], o.ECFFT_COSET[1])])
        self._h = _ecfft_lib().dvo_fftree_new(log_n, int(shifted), log_n if base_log_n is None else base_log_n, o.ECFFT_LOG_ORDER, _p(consts))
        if not self._h:
            raise ValueError("dvo_fftree_new failed")

    def close(self):
        if getattr(self, "_h", None):
            _ecfft_lib().dvo_fftree_free(self._h)
            self._h = None

    __del__ = close

    def layer(self, d: int = 0) -> np.ndarray:
        out = np.zeros((self.n >> d, 4), dtype=np.uint64)
        _ecfft_lib().dvo_fftree_layer(self._h, d, _p(out))
        return out

    def leaves(self) -> np.ndarray:
        return self.layer(0)

    def _run(self, fn, arr, n, *args):
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        assert a.shape == (n, 4), (a.shape, n)
        out = np.zeros_like(a)
        rc = fn(self._h, _p(a), *args, _p(out))
        assert rc == 0
        return out

    def extend(self, ev, sl: int = 0, to_even: bool = False) -> np.ndarray:
        return self._run(_ecfft_lib().dvo_fftree_extend, ev, (self.n >> sl) // 2, sl, int(to_even))

    def enter(self, coeffs, sl: int = 0) -> np.ndarray:
        return self._run(_ecfft_lib().dvo_fftree_enter, coeffs, self.n >> sl, sl)

    def exit(self, ev, sl: int = 0) -> np.ndarray:
        return self._run(_ecfft_lib().dvo_fftree_exit, ev, self.n >> sl, sl)

    def eval(self, coeffs, sl: int = 0) -> np.ndarray:
        """Horner at every leaf: the definition of enter"""
        return self._run(_ecfft_lib().dvo_fftree_eval, coeffs, self.n >> sl, sl)

    def matrices(self, to_even: bool, which: int) -> np.ndarray:
        """[(n/2 - 1) * 4, 4]: the layer-ordered 2x2 matrices of one direction (which = 0 decompose, 1 recombine)"""
        out = np.zeros(((self.n // 2 - 1) * 4, 4), dtype=np.uint64)
        assert _ecfft_lib().dvo_fftree_matrices(self._h, int(to_even), which, _p(out)) == 0
        return out

    def vanish_even_at(self, x: int, sl: int = 0) -> int:
        out = np.zeros(4, dtype=np.uint64)
        _ecfft_lib().dvo_fftree_vanish_even_at(self._h, _p(_limbs(x)), sl, _p(out))
        return _int(out)
