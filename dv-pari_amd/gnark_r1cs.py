"""Host mirror of src/gnark_r1cs.rs: the sparse R1CS (rows of (wire_id, coeff_id) terms over a
coefficient table), kept as three CSR matrices so it can be handed to the GPU prover unchanged."""
import ctypes as C
import struct
from dataclasses import dataclass

import numpy as np

from . import fr
from ._native import lib, check, ptr

P = fr.P


@dataclass
class Csr:
    row_ptr: np.ndarray  # uint32 [n_rows+1]
    wire: np.ndarray     # uint32 [nnz]
    coeff: np.ndarray    # uint32 [nnz]

    @staticmethod
    def from_rows(rows):
        """rows: list of lists of (wire_id, coeff_id)"""
        rp = np.zeros(len(rows) + 1, dtype=np.uint32)
        rp[1:] = np.cumsum([len(r) for r in rows])
        flat = [t for r in rows for t in r]
        w = np.array([t[0] for t in flat], dtype=np.uint32)
        c = np.array([t[1] for t in flat], dtype=np.uint32)
        return Csr(rp, w, c)

    def transpose(self, n_cols):
        """CSR of the transpose: rows = wires, 'wire' column then holds the original row index."""
        n_rows = self.row_ptr.shape[0] - 1
        rows = np.repeat(np.arange(n_rows, dtype=np.uint32), np.diff(self.row_ptr).astype(np.int64))
        order = np.argsort(self.wire, kind="stable")
        rp = np.zeros(n_cols + 1, dtype=np.uint32)
        np.add.at(rp, self.wire.astype(np.int64) + 1, 1)
        rp = np.cumsum(rp).astype(np.uint32)
        return Csr(rp, rows[order], self.coeff[order])


@dataclass
class R1CSInstance:
    """R1CSInstance, src/gnark_r1cs.rs:263-296 (rows as CSR, coefficient table canonical)."""
    num_constraints: int      # padded to a power of two
    num_public_inputs: int
    n_rows: int               # real rows
    n_wires: int
    l: Csr
    r: Csr
    o: Csr
    coeffs: np.ndarray        # uint64 [n_coeffs, 4]

    @staticmethod
    def from_rows(rows, coeffs, num_public_inputs, n_wires=None):
        """rows: list of (l_terms, r_terms, o_terms); coeffs: python ints."""
        m = 1
        while m < len(rows):
            m *= 2
        if n_wires is None:
            n_wires = 1 + max(t[0] for row in rows for part in row for t in part)
        return R1CSInstance(m, num_public_inputs, len(rows), n_wires, Csr.from_rows([r[0] for r in rows]),
                            Csr.from_rows([r[1] for r in rows]), Csr.from_rows([r[2] for r in rows]), fr.vec(coeffs))

    # ---- the SP1/gnark dump format, src/gnark_r1cs.rs:84-91,121-185 (writer mirrors :405-438) -------------
    def to_dump_bytes(self) -> bytes:
        nr = self.n_rows
        cnt = [np.diff(m.row_ptr[: nr + 1].astype(np.int64)) for m in (self.l, self.r, self.o)]
        words = 3 + 2 * (cnt[0] + cnt[1] + cnt[2])
        start = np.zeros(nr + 1, dtype=np.int64)
        start[1:] = np.cumsum(words)
        body = np.zeros(int(start[-1]), dtype="<u4")
        before = np.zeros(nr, dtype=np.int64)
        for k, m in enumerate((self.l, self.r, self.o)):
            body[start[:-1] + k] = cnt[k]
            nnz = int(m.row_ptr[nr])
            row_of = np.repeat(np.arange(nr, dtype=np.int64), cnt[k])
            local = np.arange(nnz, dtype=np.int64) - m.row_ptr[:nr].astype(np.int64)[row_of]
            dst = start[:-1][row_of] + 3 + 2 * (before[row_of] + local)
            body[dst] = m.wire[:nnz]
            body[dst + 1] = m.coeff[:nnz]
            before += cnt[k]
        coeff_be = np.ascontiguousarray(self.coeffs, dtype="<u8").view(np.uint8).reshape(-1, 32)[:, ::-1]
        return b"".join([struct.pack("<I", self.coeffs.shape[0]), coeff_be.tobytes(), struct.pack("<I", nr), body.tobytes()])

    def write_dump_file(self, path):
        with open(path, "wb") as f:
            f.write(self.to_dump_bytes())

    @staticmethod
    def from_dump_bytes(buf: bytes, num_public_inputs: int):
        """load_sparse_r1cs_from_file + from_dump, src/gnark_r1cs.rs:121-185,282-296 (native parser,
        dvp_r1cs_dump_*): n_wires = max wire id + 1 as in accumulate_m_values, src/srs.rs:56-62."""
        b = np.frombuffer(buf, dtype=np.uint8)
        nc, nr, nw = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        nnz = (C.c_uint64 * 3)()
        check(lib.dvp_r1cs_dump_sizes(ptr(b), b.shape[0], C.byref(nc), C.byref(nr), C.byref(nnz), C.byref(nw)), "dvp_r1cs_dump_sizes")
        coeffs = np.zeros((nc.value, 4), dtype=np.uint64)
        rp = [np.zeros(nr.value + 1, dtype=np.uint32) for _ in range(3)]
        wi = [np.zeros(max(int(nnz[k]), 1), dtype=np.uint32) for k in range(3)]
        ci = [np.zeros(max(int(nnz[k]), 1), dtype=np.uint32) for k in range(3)]
        arr = lambda xs: (C.c_void_p * 3)(*[x.ctypes.data for x in xs])
        a_rp, a_wi, a_ci = arr(rp), arr(wi), arr(ci)
        check(lib.dvp_r1cs_dump_fill(ptr(b), b.shape[0], ptr(coeffs), C.byref(a_rp), C.byref(a_wi), C.byref(a_ci)), "dvp_r1cs_dump_fill")
        m = 1
        while m < nr.value:
            m *= 2
        mats = [Csr(rp[k], wi[k][: int(nnz[k])], ci[k][: int(nnz[k])]) for k in range(3)]
        return R1CSInstance(m, num_public_inputs, nr.value, nw.value, mats[0], mats[1], mats[2], coeffs)

    @staticmethod
    def from_dump_file(path, num_public_inputs: int):
        with open(path, "rb") as f:
            return R1CSInstance.from_dump_bytes(f.read(), num_public_inputs)


def load_witness_bytes(buf: bytes):
    """witness file: u32-BE count || count x 32-byte BE elements (src/gnark_r1cs.rs:58-77,188-198)."""
    (n,) = struct.unpack_from(">I", buf, 0)
    return [int.from_bytes(buf[4 + 32 * i: 4 + 32 * (i + 1)], "big") % P for i in range(n)]


def load_witness_from_file(path) -> np.ndarray:
    """load_witness_from_file, src/gnark_r1cs.rs:188-210 -> uint64 [n,4] canonical (values reduced mod p)"""
    import os

    pth = os.fspath(path).encode()
    n = C.c_size_t(0)
    check(lib.dvp_file_witness_read(pth, None, 0, C.byref(n)), f"load_witness_from_file({path})")
    out = np.zeros((n.value, 4), dtype=np.uint64)
    if n.value:
        check(lib.dvp_file_witness_read(pth, ptr(out), n.value, C.byref(n)), f"load_witness_from_file({path})")
    return out


def write_witness_to_file(path, values):
    """the writer side of the witness format (the reference only reads it; used by the synthetic generators)"""
    import os

    v = values if isinstance(values, np.ndarray) else fr.vec(values)
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
    check(lib.dvp_file_witness_write(os.fspath(path).encode(), ptr(v), v.shape[0]), f"write_witness_to_file({path})")


def evaluate_monomial_basis_poly(public_inputs, alpha):
    """src/gnark_r1cs.rs:391-399"""
    acc, pw = 0, 1
    for x in public_inputs:
        acc = (acc + x * pw) % P
        pw = pw * alpha % P
    return acc


TOY_COEFFS = [1, 2]
TOY_ROWS = [  # src/dvsnark_test.rs:84-115 -- wires 1=0, o=1, w=2, y=3, z=4, x=5, t=6, s=7
    ([(5, 0)], [(5, 0)], [(3, 0)]),
    ([(3, 0), (4, 0)], [(0, 0)], [(2, 0)]),
    ([(4, 1)], [(0, 0)], [(6, 0)]),
    ([(5, 0), (6, 0)], [(0, 0)], [(7, 0)]),
    ([(2, 0), (7, 0)], [(0, 0)], [(1, 0)]),
]
TOY_PUBLIC = [24, 13]
TOY_PRIVATE = [9, 4, 3, 8, 11]


def synthetic_dense(log_m: int, seed: int = 0x5EED0004, n_coeffs: int = 256):
    """BASELINE config #4: m = 2^log_m real rows, n_wires = m, wires [1, pub0, pub1, private...]; row i is
    (w[a] + c*w[b]) * w[d] = w[o] with a fresh o while wires remain, then identity rows w[a]*1 = w[a].
    Returns (R1CSInstance, public_inputs, private_inputs), the witness built by forward evaluation."""
    m = 1 << log_m
    rng = np.random.default_rng(seed)
    coeffs = [1] + [int.from_bytes(rng.bytes(28), "little") for _ in range(n_coeffs - 1)]
    n_free = 4  # wires 0..3 are inputs: 1, pub0, pub1, priv0
    w = [1] + [int.from_bytes(rng.bytes(28), "little") for _ in range(3)] + [0] * (m - n_free)
    a_idx = rng.integers(0, 1 << 62, size=m)
    b_idx = rng.integers(0, 1 << 62, size=m)
    d_idx = rng.integers(0, 1 << 62, size=m)
    c_idx = rng.integers(0, n_coeffs, size=m)
    lw = np.zeros((m, 2), dtype=np.uint32)
    lc = np.zeros((m, 2), dtype=np.uint32)
    rw = np.zeros(m, dtype=np.uint32)
    ow = np.zeros(m, dtype=np.uint32)
    for i in range(m):
        o = n_free + i
        if o < m:
            a, b, d = int(a_idx[i]) % o, int(b_idx[i]) % o, int(d_idx[i]) % o
            ci = int(c_idx[i])
            w[o] = (w[a] + coeffs[ci] * w[b]) % P * w[d] % P
            lw[i] = (a, b)
            lc[i] = (0, ci)
            rw[i], ow[i] = d, o
        else:  # wires exhausted: w[a] * 1 = w[a]
            a = int(a_idx[i]) % m
            lw[i] = (a, a)
            lc[i] = (0, 0)
            # (w[a] + 1*w[a]) * 1 = 2 w[a] would need a new wire; use l = {a}, so make the second term vanish:
            rw[i], ow[i] = 0, a
    # rows with two L terms, except the tail rows which have one
    tail = max(0, m - (m - n_free))
    n_l = np.full(m, 2, dtype=np.int64)
    n_l[m - n_free:] = 1
    rp_l = np.zeros(m + 1, dtype=np.uint32)
    rp_l[1:] = np.cumsum(n_l)
    mask = np.ones((m, 2), dtype=bool)
    mask[m - n_free:, 1] = False
    l = Csr(rp_l, lw[mask], lc[mask])
    rp1 = np.arange(m + 1, dtype=np.uint32)
    zeros = np.zeros(m, dtype=np.uint32)
    r = Csr(rp1, rw, zeros)
    o = Csr(rp1, ow, zeros.copy())
    inst = R1CSInstance(m, 2, m, m, l, r, o, fr.vec(coeffs))
    return inst, w[1:3], w[3:]


def synthetic_sparse(log_m: int, n_wires: int = None, max_terms: int = 8, seed: int = 0x5EED0005, n_coeffs: int = 64,
                     n_public: int = 2):
    """BASELINE config #5 stand-in: SP1-like sparse R1CS in the dump format's own terms -- every row is
    (sum_k c_k w[a_k]) * (sum_k c'_k w[b_k]) = w[o] with 1..max_terms terms per side, rows < 2^log_m so
    the instance is padded.  Returns (R1CSInstance, public, private); witness by forward evaluation."""
    m = 1 << log_m
    n_rows = m - m // 8 - 3  # deliberately not a power of two: exercises the zero-row padding
    if n_wires is None:
        n_wires = n_rows + 1 + n_public + 4
    rng = np.random.default_rng(seed)
    coeffs = [1] + [int.from_bytes(rng.bytes(28), "little") for _ in range(n_coeffs - 1)]
    n_free = 1 + n_public + 4
    assert n_wires >= n_free + n_rows
    w = [1] + [int.from_bytes(rng.bytes(28), "little") for _ in range(n_free - 1)] + [0] * (n_wires - n_free)
    rows = []
    for i in range(n_rows):
        o = n_free + i
        sides = []
        vals = []
        for _ in range(2):
            k = int(rng.integers(1, max_terms + 1))
            terms = [(int(rng.integers(0, o)), int(rng.integers(0, n_coeffs))) for _ in range(k)]
            sides.append(terms)
            vals.append(sum(coeffs[c] * w[a] for a, c in terms) % P)
        w[o] = vals[0] * vals[1] % P
        rows.append((sides[0], sides[1], [(o, 0)]))
    inst = R1CSInstance.from_rows(rows, coeffs, n_public, n_wires=n_wires)
    return inst, w[1:1 + n_public], w[1 + n_public:]


def synthetic_sparse_fast(log_m: int, max_terms: int = 8, levels: int = 32, seed: int = 0x5EED0005, n_coeffs: int = 64,
                          n_public: int = 2):
    """BASELINE config #5 at its stated size: the same circuit family as synthetic_sparse -- every row is
    (sum_k c_k w[a_k]) * (sum_k c'_k w[b_k]) = w[o], 1..max_terms terms per side, fresh output wire per row,
    rows = 7/8 * 2^log_m - 3 (so the instance is padded), SP1-like sparsity (src/gnark_r1cs.rs:84-91 is the format
    it is written in) -- but built without a python loop over rows: the structure is drawn with numpy and the witness
    is evaluated level by level (rows of a level only read wires of earlier levels) with the library's own Fr vector
    ops.  Returns (R1CSInstance, public [n,4], private [n,4]) with the witness as limb arrays."""
    m = 1 << log_m
    n_rows = m - m // 8 - 3
    n_free = 1 + n_public + 4
    n_wires = n_free + n_rows
    rng = np.random.default_rng(seed)
    coeffs = fr.vec([1] + [int.from_bytes(rng.bytes(28), "little") for _ in range(n_coeffs - 1)])
    bounds = np.linspace(0, n_rows, levels + 1).astype(np.int64)
    level_of = np.searchsorted(bounds, np.arange(n_rows), side="right") - 1
    # wires a row may read: everything produced before its level started (and the free wires)
    avail = (n_free + bounds[level_of]).astype(np.int64)
    sides = []
    for _ in range(2):
        cnt = rng.integers(1, max_terms + 1, size=n_rows).astype(np.int64)
        rp = np.zeros(n_rows + 1, dtype=np.int64)
        rp[1:] = np.cumsum(cnt)
        row_of = np.repeat(np.arange(n_rows, dtype=np.int64), cnt)
        wire = (rng.integers(0, 1 << 62, size=int(rp[-1])) % avail[row_of]).astype(np.uint32)
        cid = rng.integers(0, n_coeffs, size=int(rp[-1])).astype(np.uint32)
        sides.append(Csr(rp.astype(np.uint32), wire, cid))
    out = Csr(np.arange(n_rows + 1, dtype=np.uint32), (n_free + np.arange(n_rows)).astype(np.uint32), np.zeros(n_rows, dtype=np.uint32))
    w = np.zeros((n_wires, 4), dtype=np.uint64)
    w[0, 0] = 1
    w[1:n_free] = fr.vec([int.from_bytes(rng.bytes(28), "little") for _ in range(n_free - 1)])
    for lv in range(levels):
        lo, hi = int(bounds[lv]), int(bounds[lv + 1])
        if hi == lo:
            continue
        vals = []
        for sd in sides:
            a, b = int(sd.row_ptr[lo]), int(sd.row_ptr[hi])
            rp = (sd.row_ptr[lo:hi + 1].astype(np.int64) - a).astype(np.uint32)
            vals.append(fr.spmv(rp, sd.wire[a:b], sd.coeff[a:b], coeffs, w[: n_free + lo]))
        w[n_free + lo:n_free + hi] = fr.mul(vals[0], vals[1])
    inst = R1CSInstance(m, n_public, n_rows, n_wires, sides[0], sides[1], out, coeffs)
    return inst, w[1:1 + n_public].copy(), w[1 + n_public:].copy()
