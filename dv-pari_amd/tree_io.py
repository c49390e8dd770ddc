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


def check_tree_file(path, tree) -> dict:
    """Compares a (reference-built) tree file with `tree` (ec_fft.FFTree, regenerated from src/ec_fft.rs:205-229):
    the leaves must be identical; returns {'leaves': n, 'sections': [...]} or raises ValueError."""
    leaves = read_leaves(path)
    mine = tree.leaves()
    if leaves.shape != mine.shape or not np.array_equal(leaves, mine):
        bad = -1 if leaves.shape != mine.shape else int(np.nonzero((leaves != mine).any(axis=1))[0][0])
        raise ValueError(f"{path}: leaves differ from the regenerated domain (first difference at leaf {bad})")
    return {"leaves": int(leaves.shape[0]), "sections": sections(path)}
