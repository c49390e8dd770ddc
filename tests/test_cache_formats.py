"""CPU tests of the cache_dir file formats (csrc/cache.cpp, host-only code of the C ABI): every byte layout is
restated here in plain python from the reference's description (src/io_utils.rs:1-8,42-111; src/gnark_r1cs.rs:58-91)
and compared with what the library writes / parses.  No GPU work is launched."""
import os
import random
import struct

import numpy as np
import pytest

import pyref as o
from util import to_limbs, from_limbs


def py_dump(rows, coeffs):
    """the dump layout, row by row (src/gnark_r1cs.rs:84-91)"""
    out = [struct.pack("<I", len(coeffs))] + [int(c).to_bytes(32, "big") for c in coeffs] + [struct.pack("<I", len(rows))]
    for l, r, oo in rows:
        out.append(struct.pack("<III", len(l), len(r), len(oo)))
        for part in (l, r, oo):
            for w, c in part:
                out.append(struct.pack("<II", w, c))
    return b"".join(out)


def rand_rows(rnd, n_rows, n_wires, n_coeffs):
    rows = []
    for _ in range(n_rows):
        rows.append(tuple([(rnd.randrange(n_wires), rnd.randrange(n_coeffs)) for _ in range(rnd.randrange(0, 5))] for _ in range(3)))
    return rows


def test_fr_vec_file_layout(dvp, tmp_path):
    rnd = random.Random(1)
    vals = [0, 1, o.P - 1] + [rnd.randrange(o.P) for _ in range(50)]
    path = tmp_path / "z_poly"
    dvp.io_utils.write_fr_vec_to_file(path, to_limbs(vals))
    raw = path.read_bytes()
    assert raw == struct.pack("<Q", len(vals)) + b"".join(v.to_bytes(29, "little") for v in vals)
    assert from_limbs(dvp.io_utils.read_fr_vec_from_file(path)) == vals
    # empty vector, truncated payload, short prefix, non-canonical element
    dvp.io_utils.write_fr_vec_to_file(path, np.zeros((0, 4), dtype=np.uint64))
    assert path.read_bytes() == bytes(8) and dvp.io_utils.read_fr_vec_from_file(path).shape == (0, 4)
    path.write_bytes(raw[:-1])
    with pytest.raises(dvp.DvpError) as e:
        dvp.io_utils.read_fr_vec_from_file(path)
    assert e.value.status == -6
    path.write_bytes(raw[:5])
    with pytest.raises(dvp.DvpError):
        dvp.io_utils.read_fr_vec_from_file(path)
    path.write_bytes(struct.pack("<Q", 1) + o.P.to_bytes(29, "little"))
    with pytest.raises(dvp.DvpError) as e:
        dvp.io_utils.read_fr_vec_from_file(path)
    assert e.value.status == -1
    with pytest.raises(dvp.DvpError):
        dvp.io_utils.write_fr_vec_to_file(path, to_limbs([o.P]))
    with pytest.raises(dvp.DvpError):
        dvp.io_utils.read_fr_vec_from_file(tmp_path / "missing")


def test_point_vec_file_layout(dvp, tmp_path):
    rng = np.random.default_rng(2)
    enc = rng.integers(0, 256, size=(37, 30), dtype=np.uint8)
    path = tmp_path / "g_q"
    dvp.io_utils.write_point_vec_to_file(path, enc)
    assert path.read_bytes() == struct.pack("<Q", 37) + enc.tobytes()
    assert np.array_equal(dvp.io_utils.read_point_vec_payload(path), enc)
    path.write_bytes(struct.pack("<Q", 38) + enc.tobytes())
    with pytest.raises(dvp.DvpError) as e:
        dvp.io_utils.read_point_vec_payload(path)
    assert e.value.status == -6


def test_witness_file(dvp, tmp_path):
    rnd = random.Random(3)
    vals = [0, 1, o.P - 1, o.P, o.P + 5, (1 << 256) - 1] + [rnd.randrange(1 << 256) for _ in range(40)]
    path = tmp_path / "witness_to_dvsnark"
    path.write_bytes(struct.pack(">I", len(vals)) + b"".join(v.to_bytes(32, "big") for v in vals))
    got = dvp.gnark_r1cs.load_witness_from_file(path)
    assert from_limbs(got) == [v % o.P for v in vals]  # Fr::from_be_bytes_mod_order, src/gnark_r1cs.rs:201-210
    assert dvp.gnark_r1cs.load_witness_bytes(path.read_bytes()) == [v % o.P for v in vals]
    dvp.gnark_r1cs.write_witness_to_file(path, [v % o.P for v in vals])
    assert path.read_bytes() == struct.pack(">I", len(vals)) + b"".join((v % o.P).to_bytes(32, "big") for v in vals)
    path.write_bytes(path.read_bytes()[:-3])
    with pytest.raises(dvp.DvpError):
        dvp.gnark_r1cs.load_witness_from_file(path)


def test_r1cs_dump_roundtrip(dvp):
    rnd = random.Random(4)
    coeffs = [1, 2, o.P - 1] + [rnd.randrange(o.P) for _ in range(13)]
    rows = rand_rows(rnd, 77, 40, len(coeffs))
    rows[5] = ([], [], [])  # an empty row is legal
    rows[9] = ([(39, 0)], [(0, 1)], [(3, 2)])
    raw = py_dump(rows, coeffs)
    inst = dvp.gnark_r1cs.R1CSInstance.from_dump_bytes(raw, 2)
    assert (inst.num_constraints, inst.n_rows, inst.n_wires, inst.num_public_inputs) == (128, 77, 40, 2)
    assert from_limbs(inst.coeffs) == coeffs
    for k, mt in enumerate((inst.l, inst.r, inst.o)):
        for i, row in enumerate(rows):
            a, b = int(mt.row_ptr[i]), int(mt.row_ptr[i + 1])
            assert [(int(mt.wire[j]), int(mt.coeff[j])) for j in range(a, b)] == row[k]
    assert inst.to_dump_bytes() == raw  # the vectorised writer reproduces the layout byte for byte
    ref = dvp.gnark_r1cs.R1CSInstance.from_rows(rows, coeffs, 2)
    assert ref.to_dump_bytes() == raw
    # coefficients >= p are reduced (from_be_bytes_mod_order, src/gnark_r1cs.rs:284-288)
    big = py_dump([([(0, 0)], [(0, 1)], [(0, 0)])], [o.P + 7, (1 << 256) - 1])
    assert from_limbs(dvp.gnark_r1cs.R1CSInstance.from_dump_bytes(big, 0).coeffs) == [7, ((1 << 256) - 1) % o.P]
    # malformed inputs: truncated anywhere, coefficient id out of range
    for cut in (3, 4 + 32 * len(coeffs) + 2, len(raw) - 1, len(raw) - 9):
        with pytest.raises(dvp.DvpError) as e:
            dvp.gnark_r1cs.R1CSInstance.from_dump_bytes(raw[:cut], 2)
        assert e.value.status == -6, cut
    bad = py_dump([([(0, 5)], [], [])], [1, 2])
    with pytest.raises(dvp.DvpError) as e:
        dvp.gnark_r1cs.R1CSInstance.from_dump_bytes(bad, 0)
    assert e.value.status == -1


def test_toy_dump_matches_rows(dvp):
    g = dvp.gnark_r1cs
    raw = py_dump(g.TOY_ROWS, g.TOY_COEFFS)
    inst = g.R1CSInstance.from_dump_bytes(raw, 2)
    assert inst.num_constraints == 8 and inst.n_wires == 8 and inst.n_rows == 5


def test_open_cache_dir_errors_without_gpu_work(dvp, tmp_path):
    """missing files are reported as DVP_EIO before any device work is attempted"""
    import ctypes as C

    h = C.c_void_p()
    assert dvp.lib.dvp_prover_open_cache_dir(os.fspath(tmp_path).encode(), 2, C.byref(h)) == -6
    (tmp_path / "r1cs_to_dvsnark").write_bytes(py_dump(dvp.gnark_r1cs.TOY_ROWS, dvp.gnark_r1cs.TOY_COEFFS))
    assert dvp.lib.dvp_prover_open_cache_dir(os.fspath(tmp_path).encode(), 2, C.byref(h)) == -6  # SRS files missing
    assert not h.value


def test_r1cs_dump_parser_survives_corruption(dvp):
    """random truncations and byte flips of a valid dump: the native parser either accepts (and then agrees with a plain
    python walk of the same bytes) or reports a status -- it never reads out of bounds"""
    rnd = random.Random(99)
    coeffs = [rnd.randrange(o.P) for _ in range(5)]
    rows = rand_rows(rnd, 40, 30, len(coeffs))
    good = py_dump(rows, coeffs)

    def py_walk(buf):
        """None if malformed, else (n_coeffs, n_rows, nnz)"""
        try:
            (nc,) = struct.unpack_from("<I", buf, 0)
            off = 4 + 32 * nc
            if off + 4 > len(buf):
                return None
            (nr,) = struct.unpack_from("<I", buf, off)
            off += 4
            nnz = [0, 0, 0]
            for _ in range(nr):
                cnt = struct.unpack_from("<III", buf, off)
                off += 12
                for k in range(3):
                    if off + 8 * cnt[k] > len(buf):
                        return None
                    for t in range(cnt[k]):
                        _, c = struct.unpack_from("<II", buf, off + 8 * t)
                        if c >= nc:
                            return "badcoeff"
                    nnz[k] += cnt[k]
                    off += 8 * cnt[k]
            return nc, nr, nnz
        except struct.error:
            return None

    for it in range(300):
        b = bytearray(good)
        if it % 3 == 0:
            b = b[: rnd.randrange(0, len(b))]
        else:
            for _ in range(rnd.randrange(1, 4)):
                if b:
                    b[rnd.randrange(len(b))] = rnd.randrange(256)
        b = bytes(b)
        exp = py_walk(b)
        try:
            inst = dvp.gnark_r1cs.R1CSInstance.from_dump_bytes(b, 2)
            got = (inst.coeffs.shape[0], inst.n_rows, [int(mt.row_ptr[-1]) for mt in (inst.l, inst.r, inst.o)])
        except dvp.DvpError as e:
            got = None if e.status == -6 else "badcoeff" if e.status == -1 else e.status
        if isinstance(exp, tuple) and exp[1] == 0:
            continue  # zero rows: accepted by the walk, padded to one row by the host mirror
        assert got == (exp if not isinstance(exp, tuple) else (exp[0], exp[1], exp[2])), (it, exp, got)
