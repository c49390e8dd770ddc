"""CPU tests of the FFTR container reader / writer (src/tree_io.rs:1-15,144-214,353-433; csrc/tree_io.cpp is host
code, so nothing here needs a GPU): round trip, nested subtree nodes, and a corruption fuzz -- a damaged file must be
rejected with DVP_EIO / DVP_EINVAL, never crash or return out-of-range data."""
import random
import struct

import numpy as np
import pytest

import pyref as o
from util import to_limbs, from_limbs

MAGIC = b"FFTR\0\0\0\0"


def fr_blob(vals, per=1):
    assert len(vals) % per == 0
    return struct.pack("<Q", len(vals) // per) + b"".join(int(v).to_bytes(29, "little") for v in vals)


def node_bytes(sections):
    """sections: [(id, blob bytes)] -> node laid out as write_fftree_to_vec does (src/tree_io.rs:167-214)"""
    out = struct.pack("<II", len(sections), 0)
    cur = 8 + 24 * len(sections)
    for sid, blob in sections:
        out += bytes([sid]) + bytes(7) + struct.pack("<QQ", cur, len(blob))
        cur += len(blob)
    return out + b"".join(b for _, b in sections)


def file_bytes(node):
    return MAGIC + struct.pack("<Q", len(node)) + node


def test_fftr_roundtrip_and_layout(dvp, tmp_path):
    rnd = random.Random(1)
    n = 16
    f = [0] * n + [rnd.randrange(o.P) for _ in range(n)]
    rec = [rnd.randrange(o.P) for _ in range(4 * (n - 1))]
    dec = [rnd.randrange(o.P) for _ in range(4 * (n - 1))]
    xnn = [rnd.randrange(o.P) for _ in range(n)]
    path = tmp_path / "tree"
    dvp.tree_io.write_tree_file(path, {"f": to_limbs(f), "recombine_matrices": to_limbs(rec), "decompose_matrices": to_limbs(dec), "xnn_s": to_limbs(xnn)})
    # byte-for-byte what the reference's writer lays out for the same sections
    expect = file_bytes(node_bytes([(0, fr_blob(f)), (1, fr_blob(rec, 4)), (2, fr_blob(dec, 4)), (4, fr_blob(xnn))]))
    assert path.read_bytes() == expect
    assert dvp.tree_io.sections(path) == [("f", 8 + 29 * 2 * n), ("recombine_matrices", 8 + 29 * 4 * (n - 1)),
                                          ("decompose_matrices", 8 + 29 * 4 * (n - 1)), ("xnn_s", 8 + 29 * n)]
    assert from_limbs(dvp.tree_io.read_section(path, "f")) == f
    assert from_limbs(dvp.tree_io.read_section(path, 1)) == rec
    assert from_limbs(dvp.tree_io.read_section(path, "decompose_matrices")) == dec
    assert from_limbs(dvp.tree_io.read_leaves(path)) == f[n:]
    with pytest.raises(dvp.DvpError) as e:  # "missing section"
        dvp.tree_io.read_section(path, "z0_s1")
    assert e.value.status == -1
    with pytest.raises(dvp.DvpError):        # a non-canonical element cannot be written
        dvp.tree_io.write_tree_file(tmp_path / "bad", {"f": to_limbs([o.P, 1])})


def test_fftr_nested_subtree(dvp, tmp_path):
    """section 12 = one complete child node with the same layout (src/tree_io.rs:12-15,185-187); depth walks the links"""
    inner = node_bytes([(0, fr_blob([0, 0, 7, 9]))])
    mid = node_bytes([(0, fr_blob([0] * 4 + [1, 2, 3, 4])), (12, inner)])
    top = node_bytes([(0, fr_blob([0] * 8 + list(range(10, 18)))), (4, fr_blob([5] * 8)), (12, mid)])
    path = tmp_path / "nested"
    path.write_bytes(file_bytes(top))
    assert [s[0] for s in dvp.tree_io.sections(path)] == ["f", "xnn_s", "subtree"]
    assert from_limbs(dvp.tree_io.read_leaves(path, 0)) == list(range(10, 18))
    assert from_limbs(dvp.tree_io.read_leaves(path, 1)) == [1, 2, 3, 4]
    assert from_limbs(dvp.tree_io.read_leaves(path, 2)) == [7, 9]
    with pytest.raises(dvp.DvpError) as e:
        dvp.tree_io.sections(path, 3)
    assert e.value.status == -1


def test_fftr_corruption_fuzz(dvp, tmp_path):
    rnd = random.Random(99)
    n = 8
    good = file_bytes(node_bytes([(0, fr_blob([0] * n + [rnd.randrange(o.P) for _ in range(n)])),
                                  (1, fr_blob([rnd.randrange(o.P) for _ in range(4 * (n - 1))], 4)),
                                  (2, fr_blob([rnd.randrange(o.P) for _ in range(4 * (n - 1))], 4))]))
    path = tmp_path / "fuzz"
    rejected = 0
    cases = [good[:k] for k in (0, 7, 15, 16, 23, 40, 100, len(good) - 1)]  # truncations
    cases += [b"FFTX" + good[4:], good[:8] + struct.pack("<Q", len(good)) + good[16:]]  # magic, total too large
    for _ in range(300):
        b = bytearray(good)
        for _ in range(rnd.randrange(1, 4)):
            b[rnd.randrange(len(b))] = rnd.randrange(256)
        cases.append(bytes(b))
    n_struct = 10  # the first ten cases damage the structure and must all be rejected
    for idx, c in enumerate(cases):
        path.write_bytes(c)
        before = rejected
        try:
            secs = dvp.tree_io.sections(path)
            for name, _ in secs:
                if name not in ("rational_maps", "subtree"):
                    v = dvp.tree_io.read_section(path, name)
                    assert all(x < o.P for x in from_limbs(v))   # whatever is returned is canonical
        except dvp.DvpError as e:
            assert e.status in (-1, -6)
            rejected += 1
        assert idx >= n_struct or rejected == before + 1, idx
    assert rejected >= 25  # header / table / count / range damage is caught; payload damage below p is data, not structure
