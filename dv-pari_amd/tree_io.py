"""Host mirror of src/tree_io.rs: the sectioned FFTR container of the reference's FFTree files (tree2n, tree2nd, treen,
treend; src/artifacts.rs:30-57).  The GPU prover regenerates its twiddles from the curve constants and reads none of
them; this module CHECKS a reference-built tree file against the regenerated domain and writes files in the same
container.  Parsing is native (csrc/tree_io.cpp), which also states what is assumed about the third-party blob layout."""
import ctypes as C
import os

import numpy as np

from ._native import lib, check, ptr

SECTION_NAMES = ("f", "recombine_matrices", "decompose_matrices", "rational_maps", "xnn_s", "xnn_s_inv", "z0_s1", "z1_s0",
                 "z0_inv_s1", "z1_inv_s0", "z0z0_rem_xnn_s", "z1z1_rem_xnn_s", "subtree")  # src/tree_io.rs:32-48


def sections(path, depth: int = 0):
    """[(section name, blob bytes)] of the node `depth` subtree links below the root"""
    ids = np.zeros(13, dtype=np.uint8)
    lens = np.zeros(13, dtype=np.uint64)
    n = C.c_uint32(0)
    check(lib.dvp_fftr_sections(os.fspath(path).encode(), depth, ptr(ids), ptr(lens), C.byref(n)), f"fftr sections({path})")
    return [(SECTION_NAMES[int(ids[k])], int(lens[k])) for k in range(n.value)]


def read_section(path, section, depth: int = 0) -> np.ndarray:
    """field elements of a section as uint64 [n,4] canonical limbs (a matrix section: [n_matrices*4, 4], row-major)"""
    sid = SECTION_NAMES.index(section) if isinstance(section, str) else int(section)
    n = C.c_size_t(0)
    pth = os.fspath(path).encode()
    check(lib.dvp_fftr_read_fr(pth, depth, sid, None, 0, C.byref(n)), f"fftr read({path}, {section})")
    out = np.zeros((n.value, 4), dtype=np.uint64)
    if n.value:
        check(lib.dvp_fftr_read_fr(pth, depth, sid, ptr(out), n.value, C.byref(n)), f"fftr read({path}, {section})")
    return out


def read_leaves(path, depth: int = 0) -> np.ndarray:
    """FFTree::f.leaves() (src/ec_fft.rs:179-189): the second half of section 0"""
    f = read_section(path, 0, depth)
    assert f.shape[0] >= 2 and f.shape[0] % 2 == 0, "f must hold 2n elements"
    return f[f.shape[0] // 2:]


def write_tree_file(path, named_sections: dict):
    """one-node FFTR file: {section name or id: uint64 [n,4] canonical limbs}"""
    items = [(SECTION_NAMES.index(k) if isinstance(k, str) else int(k), np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
             for k, v in named_sections.items()]
    ids = np.array([i for i, _ in items], dtype=np.uint8)
    elems = np.array([v.shape[0] for _, v in items], dtype=np.uint64)
    ptrs = (C.c_void_p * len(items))(*[v.ctypes.data for _, v in items])
    check(lib.dvp_fftr_write(os.fspath(path).encode(), len(items), ptr(ids), C.cast(ptrs, C.c_void_p), ptr(elems)), f"fftr write({path})")


# ---- sections 0-2 as FFTree holds them: what read_minimal_fftree_from_file loads (src/tree_io.rs:353-433) ---------------
# BinaryTree<T> (ecfft::utils) is a heap-ordered Vec: entry 0 is padding, the root is entry 1 and a layer of k entries
# sits at [k, 2k) -- the leaves are the second half, which is how the reference reads them (f.leaves(),
# src/ec_fft.rs:179-189).  For an N-leaf tree:
#   f                   2N field elements: layer d of the isogeny chain (N >> d values) at [N >> d, 2 (N >> d))
#   recombine_matrices  N Mat2x2: layer d holds one matrix per pair (L_d[j], L_d[j + (N >> d) / 2]), j < (N >> d) / 2, at
#                       [(N >> d) / 2, N >> d):  [[v0, s0 v0], [v1, s1 v1]], v = (s - x0_d)^((N >> d) / 4 - 1)  (Lemma 3.2)
#   decompose_matrices  the inverses, same positions
# Even j are the pairs of even leaves -- extend(.., Moiety::S1) decomposes with them and the mirrored direction
# recombines with them -- odd j the pairs of odd leaves.  The last layer (two leaves, one pair) and entry 0 stay
# Mat2x2::identity.  The layout of BinaryTree / Mat2x2 and the normalisation of the isogeny's denominator (monic here)
# are third-party (alpenlabs/ecfft@9c6cac7, not in the reference tree): restated, not pinned by a reference-written file.
def _fr_one_rows(k):
    a = np.zeros((k, 4), dtype=np.uint64)
    a[:, 0] = 1
    return a


def tree_sections(tree) -> dict:
    """{'f', 'recombine_matrices', 'decompose_matrices'} of `tree` (ec_fft.FFTree) in the reference's layout"""
    N, log_n = tree.n, tree.log_n
    f = np.zeros((2 * N, 4), dtype=np.uint64)
    for d in range(log_n + 1):
        k = N >> d
        lay = np.zeros((k, 4), dtype=np.uint64)
        check(lib.dvp_debug_ecfft_layer(tree._h, d, ptr(lay)), "dvp_debug_ecfft_layer")
        f[k:2 * k] = lay
    ident = np.zeros((4, 4), dtype=np.uint64)
    ident[0, 0] = ident[3, 0] = 1
    rec = np.tile(ident, (N, 1)).reshape(N, 4, 4)
    dec = rec.copy()
    if log_n >= 2:
        n = N // 2
        got = {}
        for to_even in (0, 1):
            for which in (0, 1):
                a = np.zeros(((n - 1) * 4, 4), dtype=np.uint64)
                check(lib.dvp_debug_ecfft_matrices(tree._h, to_even, which, ptr(a)), "dvp_debug_ecfft_matrices")
                got[(to_even, which)] = a.reshape(n - 1, 4, 4)
        for d in range(log_n - 1):
            nd, off = n >> d, n - (n >> d)  # nd pairs in this layer, nd / 2 per parity
            sl = slice(off, off + nd // 2)
            dec[nd:2 * nd:2] = got[(0, 0)][sl]      # even pairs: the decompose step of extend(S1)
            dec[nd + 1:2 * nd:2] = got[(1, 0)][sl]  # odd pairs: the decompose step of the mirrored extend
            rec[nd:2 * nd:2] = got[(1, 1)][sl]      # even pairs recombine onto the even leaves
            rec[nd + 1:2 * nd:2] = got[(0, 1)][sl]
    return {"f": f, "recombine_matrices": rec.reshape(4 * N, 4), "decompose_matrices": dec.reshape(4 * N, 4)}


def write_minimal_tree_file(path, tree):
    """the three sections FFTree::extend needs (src/tree_io.rs:353-433, test :481-502), from the regenerated tree
    (assembled natively: dvp_ecfft_write_tree_file; tree_sections above is the same layout in numpy, kept as its check)"""
    check(lib.dvp_ecfft_write_tree_file(tree._h, os.fspath(path).encode()), f"dvp_ecfft_write_tree_file({path})")


def check_tree_file_native(path, tree, matrices: bool = False):
    """dvp_ecfft_check_tree_file: the comparison of check_tree_file below inside the library (what
    dvp_prover_prepares_precomputes runs on a tree2n it finds); raises ValueError with the first difference"""
    sec, ent = C.c_int(-1), C.c_int64(-1)
    rc = lib.dvp_ecfft_check_tree_file(tree._h, os.fspath(path).encode(), int(matrices), C.byref(sec), C.byref(ent))
    if rc == -1 and sec.value >= 0:  # DVP_EINVAL with a section: a difference, not a bad argument
        raise ValueError(f"{path}: section {SECTION_NAMES[sec.value]} differs from the regenerated tree (first difference at entry {ent.value})")
    check(rc, f"dvp_ecfft_check_tree_file({path})")


def check_tree_file(path, tree, matrices: bool = False) -> dict:
    """Compares a (reference-built) tree file with `tree` (ec_fft.FFTree, regenerated from src/ec_fft.rs:205-229):
    the leaves must be identical -- and, with matrices=True, the inner layers of f and both matrix sections, entry for
    entry in the layout above; returns {'leaves': n, 'sections': [...]} or raises ValueError."""
    leaves = read_leaves(path)
    mine = tree.leaves()
    if leaves.shape != mine.shape or not np.array_equal(leaves, mine):
        bad = -1 if leaves.shape != mine.shape else int(np.nonzero((leaves != mine).any(axis=1))[0][0])
        raise ValueError(f"{path}: leaves differ from the regenerated domain (first difference at leaf {bad})")
    if matrices:
        want = tree_sections(tree)
        for name in ("f", "recombine_matrices", "decompose_matrices"):
            got = read_section(path, name)
            w = want[name]
            if name == "f":  # entry 0 is padding
                got, w = got[1:], w[1:]
            if got.shape != w.shape:
                raise ValueError(f"{path}: section {name} holds {got.shape[0]} elements, the regenerated tree {w.shape[0]}")
            if not np.array_equal(got, w):
                bad = int(np.nonzero((got != w).any(axis=1))[0][0])
                per = 1 if name == "f" else 4
                raise ValueError(f"{path}: section {name} differs from the regenerated tree (first difference at entry {(bad + (1 if name == 'f' else 0)) // per})")
    return {"leaves": int(leaves.shape[0]), "sections": sections(path)}
