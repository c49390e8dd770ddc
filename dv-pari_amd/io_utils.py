"""Host mirror of src/io_utils.rs: vectors of Fr / CurvePoint as `u64-LE count || payload` files
(29-byte canonical LE field elements, 30-byte xsk233 encodings).  Parsing is native (csrc/cache.cpp)."""
import ctypes as C
import os

import numpy as np

from ._native import lib, check, ptr


def _p(path) -> bytes:
    return os.fspath(path).encode()


def write_fr_vec_to_file(path, values: np.ndarray):
    """write_fr_vec_to_file, src/io_utils.rs:42-70; values uint64 [n,4] canonical"""
    v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
    check(lib.dvp_file_fr_vec_write(_p(path), ptr(v), v.shape[0]), f"write_fr_vec_to_file({path})")


def read_fr_vec_from_file(path) -> np.ndarray:
    """read_fr_vec_from_file, src/io_utils.rs:122-179"""
    n = C.c_size_t(0)
    check(lib.dvp_file_fr_vec_read(_p(path), None, 0, C.byref(n)), f"read_fr_vec_from_file({path})")
    out = np.zeros((n.value, 4), dtype=np.uint64)
    if n.value:
        check(lib.dvp_file_fr_vec_read(_p(path), ptr(out), n.value, C.byref(n)), f"read_fr_vec_from_file({path})")
    return out


def write_point_vec_to_file(path, enc: np.ndarray):
    """write_point_vec_to_file, src/io_utils.rs:83-111; enc uint8 [n,30] (curve.to_bytes / dvp_mulgen_batch output)"""
    e = np.ascontiguousarray(enc, dtype=np.uint8).reshape(-1, 30)
    check(lib.dvp_file_point_vec_write(_p(path), ptr(e), e.shape[0]), f"write_point_vec_to_file({path})")


def read_point_vec_payload(path) -> np.ndarray:
    """the n x 30 payload of a point-vector file, undecoded"""
    n = C.c_size_t(0)
    check(lib.dvp_file_point_vec_read(_p(path), None, 0, C.byref(n)), f"read_point_vec_from_file({path})")
    out = np.zeros((n.value, 30), dtype=np.uint8)
    if n.value:
        check(lib.dvp_file_point_vec_read(_p(path), ptr(out), n.value, C.byref(n)), f"read_point_vec_from_file({path})")
    return out


def read_point_vec_from_file(path):
    """read_point_vec_from_file, src/io_utils.rs:187-239: decoded on the GPU, (xy [n,8], inf [n]); an invalid
    encoding raises DvpError(DVP_EDECODE) where the reference asserts (:223)."""
    from . import curve

    enc = read_point_vec_payload(path)
    if enc.shape[0] == 0:
        return np.zeros((0, 8), dtype=np.uint64), np.zeros(0, dtype=np.uint8)
    return curve.from_bytes(enc)
