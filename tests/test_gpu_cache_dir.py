"""GPU tests (-m gpu) of the cache_dir drop-in: setup writes the reference's files, Proof::prove
reads them back with the reference's own signature; vanishing-polynomial / barycentric precomputes on the GPU."""
import json
import os
import random
import struct

import numpy as np
import pytest

import pyref as o
from util import to_limbs, from_limbs

pytestmark = pytest.mark.gpu
VEC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))


def poly_from_roots(roots):
    c = [1]
    for r in roots:
        n = [0] * (len(c) + 1)
        for i, v in enumerate(c):
            n[i + 1] = (n[i + 1] + v) % o.P
            n[i] = (n[i] - r * v) % o.P
        c = n
    return c


def horner(c, x):
    acc = 0
    for v in reversed(c):
        acc = (acc * x + v) % o.P
    return acc


@pytest.mark.parametrize("log_m", [1, 3, 6])
def test_vanishing_polynomial_vs_product(dvp, log_m):
    """compute_vanishing_polynomial (src/ec_fft.rs:241-282) == prod (X - s_i), the reference's own test
    (src/ec_fft.rs:820-880); barycentric weights == 1 / Z'(s_i) (src/ec_fft.rs:284-335)"""
    t = dvp.ec_fft.FFTree(2 << log_m)
    d, d2 = (from_limbs(x) for x in t.get_both_domains())
    for which, dom, other in ((0, d, d2), (1, d2, d)):
        z = from_limbs(dvp.ec_fft.compute_vanishing_polynomial(t, which))
        assert z == poly_from_roots(dom)
        bar, zinv = (from_limbs(x) for x in t.domain_tables(which))
        dz = [(i * c) % o.P for i, c in enumerate(z)][1:]
        assert bar == [pow(horner(dz, s), o.P - 2, o.P) for s in dom]
        assert zinv == [pow(horner(z, s), o.P - 2, o.P) for s in other]
    t.close()


def test_vanishing_polynomial_2_12_properties(dvp):
    m = 1 << 12
    t = dvp.ec_fft.FFTree(2 * m)
    rnd = random.Random(12)
    for which in (0, 1):
        z = dvp.ec_fft.compute_vanishing_polynomial(t, which)  # monic / degree checked inside
        zi = from_limbs(z)
        for _ in range(3):
            x = rnd.randrange(o.P)
            assert horner(zi, x) == t.vanish_at(which, x)
        # evaluate_vanishing_poly_at_domain (src/ec_fft.rs:407-419): zero on its own domain via enter on the 2m tree
        co = np.zeros((2 * m, 4), dtype=np.uint64)
        co[: m + 1] = z
        ev = t.enter(co)
        assert not ev[which::2].any()
        _, zinv = t.domain_tables(which)
        assert np.array_equal(dvp.fr.mul(ev[(1 - which)::2], zinv), to_limbs([1] * m))
    t.close()


def py_dump(rows, coeffs):
    out = [struct.pack("<I", len(coeffs))] + [int(c).to_bytes(32, "big") for c in coeffs] + [struct.pack("<I", len(rows))]
    for l, r, oo in rows:
        out.append(struct.pack("<III", len(l), len(r), len(oo)))
        for part in (l, r, oo):
            for w, c in part:
                out.append(struct.pack("<II", w, c))
    return b"".join(out)


def test_toy_cache_dir_end_to_end(dvp, tmp_path):
    """test_dvsnark_prover_over_toy_r1cs (src/dvsnark_test.rs:131-180) through files: dump -> verifier_runs_setup
    (writes the SRS) -> Proof::prove(cache_dir, public, private) -> verify; bytes equal the golden toy proof."""
    A, g = dvp.artifacts, dvp.gnark_r1cs
    toy = VEC["toy"]
    cache = tmp_path / "cache"
    cache.mkdir()
    (cache / A.R1CS_CONSTRAINTS_FILE).write_bytes(py_dump(g.TOY_ROWS, g.TOY_COEFFS))
    td = dvp.srs.Trapdoor(*(int(x, 16) for x in toy["trapdoor"]))
    inst, pv = dvp.srs.verifier_runs_setup_cache_dir(td, cache, 2)
    assert inst.num_constraints == 8
    for name, n in zip(A.SRS_FILES, (8, 8, 8, 8, 16)):
        raw = (cache / name).read_bytes()
        assert struct.unpack("<Q", raw[:8])[0] == n and len(raw) == 8 + 30 * n
    for name, n in ((A.Z_POLY, 9), (A.Z_POLYD, 9), (A.BAR_WTS, 8), (A.BAR_WTSD, 8), (A.Z_VALS2_INV, 8), (A.Z_VALS2D_INV, 8)):
        assert len((cache / name).read_bytes()) == 8 + 29 * n
    # the precompute files hold what the oracle's brute-force setup computes
    tree = o.FFTree(4)
    tabs = o.domain_tables(tree) if hasattr(o, "domain_tables") else None
    if tabs:
        assert from_limbs(dvp.io_utils.read_fr_vec_from_file(cache / A.BAR_WTS)) == tabs["bar_wts"]
        assert from_limbs(dvp.io_utils.read_fr_vec_from_file(cache / A.Z_VALS2_INV)) == tabs["z_vals2inv"]
    proof = dvp.proving.Proof.prove(cache, g.TOY_PUBLIC, g.TOY_PRIVATE)
    assert proof.commit_p.hex() == toy["commit_p"] and proof.kzg_k.hex() == toy["kzg_k"]
    assert proof == pv.prove(g.TOY_PUBLIC, g.TOY_PRIVATE)
    assert dvp.srs.verify(td, g.TOY_PUBLIC, proof)
    # second call reuses the opened prover; an unsatisfied witness is reported with its row
    assert dvp.proving.Proof.prove(cache, g.TOY_PUBLIC, g.TOY_PRIVATE) == proof
    bad = list(g.TOY_PRIVATE)
    bad[3] += 1
    with pytest.raises(dvp.DvpError) as e:
        dvp.proving.Proof.prove(cache, g.TOY_PUBLIC, bad)
    assert e.value.status == -3 and e.value.index == 2
    # prover_prepares_precomputes (src/proving.rs:225-325, the C entry) regenerates the same files from z_poly alone, writes the
    # minimal tree2n, and validates z_poly and whatever it finds
    P = dvp.proving
    before = {n: (cache / n).read_bytes() for n in (A.BAR_WTS, A.Z_VALS2_INV)}
    for n in before:
        os.remove(cache / n)
    assert not (cache / A.TREE_2N).exists()
    rep = P.prover_prepares_precomputes(cache, validate_precompute=True)
    assert rep == P.PREP_WROTE_TREE2N | P.PREP_WROTE_BAR_WTS | P.PREP_WROTE_Z_VALS2INV
    assert {n: (cache / n).read_bytes() for n in before} == before
    t16 = dvp.ec_fft.FFTree(16)
    dvp.tree_io.check_tree_file(cache / A.TREE_2N, t16, matrices=True)  # the numpy restatement of the layout agrees
    t16.close()
    assert P.prover_prepares_precomputes(cache, validate_precompute=True) == 0  # everything found, everything validated
    os.remove(cache / A.Z_VALS2_INV)
    assert P.prover_prepares_precomputes(cache) == P.PREP_WROTE_Z_VALS2INV
    assert (cache / A.Z_VALS2_INV).read_bytes() == before[A.Z_VALS2_INV]
    # a damaged bar_wts / tree2n is found by validation only (the reference reads them unchecked)
    bw = dvp.io_utils.read_fr_vec_from_file(cache / A.BAR_WTS)
    bw[5, 0] ^= 1
    dvp.io_utils.write_fr_vec_to_file(cache / A.BAR_WTS, bw)
    assert P.prover_prepares_precomputes(cache) == 0
    with pytest.raises(ValueError, match="bar_wts"):
        P.prover_prepares_precomputes(cache, validate_precompute=True)
    (cache / A.BAR_WTS).write_bytes(before[A.BAR_WTS])
    good_tree = (cache / A.TREE_2N).read_bytes()
    raw = bytearray(good_tree)
    raw[-40] ^= 1  # inside the last matrix blob
    (cache / A.TREE_2N).write_bytes(bytes(raw))
    with pytest.raises((ValueError, dvp.DvpError)):
        P.prover_prepares_precomputes(cache, validate_precompute=True)
    (cache / A.TREE_2N).write_bytes(good_tree)
    # z_poly: c * Z_D still vanishes on D (passes, flagged as not monic); one changed coefficient does not; all zero is refused;
    # a missing file is an I/O error
    z = dvp.io_utils.read_fr_vec_from_file(cache / A.Z_POLY)
    z2 = to_limbs([(2 * v) % o.P for v in from_limbs(z)])
    dvp.io_utils.write_fr_vec_to_file(cache / A.Z_POLY, z2)
    # ... and the tables the reference derives from THAT z_poly (1 / Z'(d_i), 1 / Z(d'_i), src/proving.rs:284-304) are the monic ones
    # times 1 / 2: the monic files now in the directory no longer belong to it, the regenerated ones do
    with pytest.raises(ValueError, match="bar_wts"):
        P.prover_prepares_precomputes(cache, validate_precompute=True)
    monic = {n: from_limbs(dvp.io_utils.read_fr_vec_from_file(cache / n)) for n in (A.BAR_WTS, A.Z_VALS2_INV)}
    os.remove(cache / A.BAR_WTS)
    os.remove(cache / A.Z_VALS2_INV)
    assert P.prover_prepares_precomputes(cache) == P.PREP_WROTE_BAR_WTS | P.PREP_WROTE_Z_VALS2INV | P.PREP_Z_POLY_NOT_MONIC
    half = pow(2, -1, o.P)
    for n in (A.BAR_WTS, A.Z_VALS2_INV):
        assert from_limbs(dvp.io_utils.read_fr_vec_from_file(cache / n)) == [v * half % o.P for v in monic[n]]
    assert P.prover_prepares_precomputes(cache, validate_precompute=True) == P.PREP_Z_POLY_NOT_MONIC
    for n in (A.BAR_WTS, A.Z_VALS2_INV):
        (cache / n).write_bytes(before[n])
    zb = z.copy()
    zb[0, 0] ^= 1
    dvp.io_utils.write_fr_vec_to_file(cache / A.Z_POLY, zb)
    assert P.prover_prepares_precomputes(cache) == 0  # not validated: accepted, as in the reference
    with pytest.raises(ValueError, match="vanishing poly does not evaluate to zero"):
        P.prover_prepares_precomputes(cache, validate_precompute=True)
    dvp.io_utils.write_fr_vec_to_file(cache / A.Z_POLY, np.zeros_like(z))
    with pytest.raises(ValueError, match="z_poly"):
        P.prover_prepares_precomputes(cache, validate_precompute=True)
    os.remove(cache / A.Z_POLY)
    with pytest.raises(dvp.DvpError) as e:
        P.prover_prepares_precomputes(cache)
    assert e.value.status == -6  # DVP_EIO
    dvp.io_utils.write_fr_vec_to_file(cache / A.Z_POLY, z)
    assert P.prover_prepares_precomputes(cache, validate_precompute=True) == 0
    # a corrupted SRS point is refused at load time (assert!(valid), src/io_utils.rs:223)
    dvp.proving.release_cache_dir(cache)
    raw = bytearray((cache / A.SRS_G_Q).read_bytes())
    good = bytes(raw)
    hit = False
    for probe in range(256):  # find a byte value that makes point 3 undecodable
        raw[8 + 30 * 3 + 1] = probe
        (cache / A.SRS_G_Q).write_bytes(bytes(raw))
        try:
            dvp.proving.Proof.prove(cache, g.TOY_PUBLIC, g.TOY_PRIVATE)
            dvp.proving.release_cache_dir(cache)
        except dvp.DvpError as err:
            assert err.status == -2 and err.index == 3
            hit = True
            break
    assert hit
    (cache / A.SRS_G_Q).write_bytes(good)
    assert dvp.proving.Proof.prove(cache, g.TOY_PUBLIC, g.TOY_PRIVATE) == proof
    dvp.proving.release_cache_dir()
    pv.close()


@pytest.mark.parametrize("shape", ["toy", "sparse7", "sparse12"])
def test_setup_cache_dir_entry_scalars(dvp, tmp_path, shape):
    """dvp_setup_cache_dir (csrc/setup.hip: SRS::verifier_runs_setup behind the C ABI, src/srs.rs:177-361) against (i) the
    oracle's brute-force setup (domain tables by products over the whole domain, accumulate_m_values row by row) where the
    big-int oracle can go, (ii) the python orchestration over the per-operation seams (dvp.srs.srs_scalars_hostside, rounds 1-4's
    route, which the GPU tests of the prover pinned to the oracle) and (iii) the in-memory entry of the same device pipeline
    (dvp.srs.srs_scalars -> dvp_setup_scalars); then the files: every point file decodes to scalar x G."""
    A, g = dvp.artifacts, dvp.gnark_r1cs
    cache = tmp_path / "c"
    cache.mkdir()
    if shape == "toy":
        (cache / A.R1CS_CONSTRAINTS_FILE).write_bytes(py_dump(g.TOY_ROWS, g.TOY_COEFFS))
        n_pub = 2
    else:
        inst0, pub, _ = g.synthetic_sparse(int(shape[6:]))
        inst0.write_dump_file(cache / A.R1CS_CONSTRAINTS_FILE)
        n_pub = len(pub)
    rnd = random.Random(len(shape))
    trap = (rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    td = dvp.srs.Trapdoor(*trap)
    inst, pv, (g_m, g_q, g_k) = dvp.srs.verifier_runs_setup_cache_dir(td, cache, n_pub, write_precomputes=True, return_scalars=True)
    m = inst.num_constraints
    for route in (dvp.srs.srs_scalars_hostside, dvp.srs.srs_scalars):
        p_gm, p_gq, p_gk = route(pv, inst, td)
        assert np.array_equal(g_m, p_gm) and np.array_equal(g_q, p_gq), route.__name__
        for j in range(3):
            assert np.array_equal(g_k[j], p_gk[j]), (route.__name__, j)
    if m <= 128:
        rows = []
        for i in range(inst.n_rows):
            row = []
            for mt in (inst.l, inst.r, inst.o):
                a, b = int(mt.row_ptr[i]), int(mt.row_ptr[i + 1])
                row.append([(int(mt.wire[k]), int(mt.coeff[k])) for k in range(a, b)])
            rows.append(tuple(row))
        st = o.setup_srs_scalars(o.FFTree(pv.log_m + 1), rows, from_limbs(inst.coeffs), n_pub, trap)
        assert from_limbs(g_m) == st["g_m"] + [0] * (inst.n_wires - len(st["g_m"]))
        assert from_limbs(g_q) == st["g_q"]
        for j in range(3):
            assert from_limbs(g_k[j]) == st["g_k"][j]
        tb = st["tables"]
        for name, key in ((A.BAR_WTS, "bar_wts"), (A.BAR_WTSD, "bar_wtsd"), (A.Z_VALS2_INV, "z_vals2inv"), (A.Z_VALS2D_INV, "z_vals2dinv"),
                          (A.Z_POLY, "z_poly"), (A.Z_POLYD, "z_polyd")):
            assert from_limbs(dvp.io_utils.read_fr_vec_from_file(cache / name)) == tb[key], name
    # the point files hold scalar x generator, in the library's codec
    for name, sc in zip(A.SRS_FILES, (g_m, g_q, g_k[0], g_k[1], g_k[2])):
        enc = dvp.io_utils.read_point_vec_payload(cache / name)
        assert np.array_equal(enc, dvp.curve.point_scalar_mul_gen_batch_bytes(sc)), name
    # zero / non-canonical trapdoor entries are refused (src/srs.rs:199-201)
    import ctypes as C
    z = np.zeros(4, dtype=np.uint64)
    t, d = dvp.fr.limbs(trap[0]), dvp.fr.limbs(trap[1])
    from importlib import import_module
    nat = import_module("dv-pari_amd._native")
    assert dvp.lib.dvp_setup_cache_dir(nat.ptr(t), nat.ptr(d), nat.ptr(z), os.fsencode(str(cache)), n_pub, 0) == -1
    pv.close()


def test_sparse_cache_dir_with_witness_file(dvp, tmp_path):
    """BASELINE config #5 stand-in at 2^12: SP1-format dump + witness file in, proof out, verifier accepts; the
    Horner and the extend route for i(X) on D' give the same bytes."""
    A, g = dvp.artifacts, dvp.gnark_r1cs
    inst0, pub, prv = g.synthetic_sparse(12)
    cache = tmp_path / "c5"
    cache.mkdir()
    inst0.write_dump_file(cache / A.R1CS_CONSTRAINTS_FILE)
    g.write_witness_to_file(cache / A.R1CS_WITNESS_FILE, [1] + pub + prv)
    rnd = random.Random(5)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    inst, pv = dvp.srs.verifier_runs_setup_cache_dir(td, cache, len(pub), write_precomputes=False)
    assert inst.n_rows == inst0.n_rows and inst.num_constraints == 1 << 12
    w = g.load_witness_from_file(cache / A.R1CS_WITNESS_FILE)
    assert from_limbs(w[:1]) == [1]
    wpub, wprv = w[1:1 + len(pub)], w[1 + len(pub):]
    # the dump knows wires only up to the highest one used; g_m has that many entries and so must the witness
    n_wires = struct.unpack("<Q", (cache / A.SRS_G_M).read_bytes()[:8])[0]
    assert n_wires == inst.n_wires <= w.shape[0]
    wprv = wprv[: n_wires - 1 - len(pub)]
    proof = dvp.proving.Proof.prove(cache, wpub, wprv)
    assert dvp.srs.verify(td, pub, proof)
    dvp.proving.release_cache_dir(cache)
    with dvp.tune(DVP_HORNER_MAX_PUB=0):  # i(X) on D' through a fourth extend instead of Horner
        assert dvp.proving.Proof.prove(cache, wpub, wprv) == proof
    dvp.proving.release_cache_dir()
    # wrong witness length: |assignment| != |g_m| (assert_eq!, src/curve.rs:142)
    with pytest.raises(dvp.DvpError) as e:
        dvp.proving.Proof.prove(cache, wpub, wprv[:-1])
    assert e.value.status == -1
    dvp.proving.release_cache_dir()
    pv.close()


@pytest.mark.parametrize("replicas", [2, 1])
def test_cache_dir_two_callers_at_once(dvp, tmp_path, replicas):
    """Proof::prove(cache_dir, ..) from three host threads at once on ONE cache_dir (cache.cpp: the second caller proves on
    a second prover opened from the same files, a third waits; DVP_CACHE_REPLICAS = 1: all take turns on one prover):
    every proof equals the bytes a single caller gets"""
    import threading

    A, g = dvp.artifacts, dvp.gnark_r1cs
    inst0, pub, prv = g.synthetic_sparse(13)
    cache = tmp_path / "twice"
    cache.mkdir()
    inst0.write_dump_file(cache / A.R1CS_CONSTRAINTS_FILE)
    rnd = random.Random(55)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    inst, pv = dvp.srs.verifier_runs_setup_cache_dir(td, cache, len(pub), write_precomputes=False)
    n_wires = struct.unpack("<Q", (cache / A.SRS_G_M).read_bytes()[:8])[0]
    w = dvp.fr.vec([1] + pub + prv)
    wpub, wprv = w[1:1 + len(pub)], w[1 + len(pub):n_wires]
    with dvp.tune(DVP_CACHE_REPLICAS=replicas):
        ref = dvp.proving.Proof.prove(cache, wpub, wprv)
        assert dvp.srs.verify(td, pub, ref)
        bad = []

        def loop():
            try:
                for _ in range(5):
                    if dvp.proving.Proof.prove(cache, wpub, wprv) != ref:
                        bad.append("bytes")
            except Exception as e:  # noqa: BLE001 -- report from the main thread
                bad.append(repr(e))

        th = [threading.Thread(target=loop) for _ in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not bad, bad
    dvp.proving.release_cache_dir()
    pv.close()


def test_cpp_host_over_the_c_abi(dvp, tmp_path):
    """examples/dvp_prove_cli.cpp -- a compiled host that sees only include/dvpari.h -- proves from a cache_dir written
    by the Python setup and prints the same 118 bytes as the Python host"""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++ on this box")
    exe = tmp_path / "dvp_prove_cli"
    libdir = os.path.join(root, "dv-pari_amd")
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "dvp_prove_cli.cpp"),
                           "-L" + libdir, "-ldvpari_hip", "-Wl,-rpath," + libdir, "-pthread", "-o", str(exe)])
    A, g = dvp.artifacts, dvp.gnark_r1cs
    inst0, pub, prv = g.synthetic_sparse(10)
    cache = tmp_path / "cache"
    cache.mkdir()
    inst0.write_dump_file(cache / A.R1CS_CONSTRAINTS_FILE)
    g.write_witness_to_file(cache / A.R1CS_WITNESS_FILE, [1] + pub + prv)
    rnd = random.Random(31)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    _, pv = dvp.srs.verifier_runs_setup_cache_dir(td, cache, len(pub), write_precomputes=False)
    ref = pv.prove(pub, prv)
    pv.close()
    env = dict(os.environ, DVP_NO_TORCH_PRELOAD="1")
    out = subprocess.run([str(exe), str(cache), str(len(pub)), str(tmp_path / "proof.bin")], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    assert bytes.fromhex(out.stdout.strip()) == ref.to_bytes()
    assert (tmp_path / "proof.bin").read_bytes() == ref.to_bytes()
    assert dvp.srs.verify(td, pub, dvp.proving.Proof.from_bytes((tmp_path / "proof.bin").read_bytes()))
    # in-library multi-GPU (dvp_set_devices) from the compiled host: four shards on device 0, identical bytes
    out = subprocess.run([str(exe), str(cache), str(len(pub)), "--devices", "0,0,0,0"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    assert bytes.fromhex(out.stdout.strip()) == ref.to_bytes()
    # the reference's own signature from two host threads of the compiled host: 12 more proofs, all the same bytes
    out = subprocess.run([str(exe), str(cache), str(len(pub)), "--repeat", "12", "--threads", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    assert bytes.fromhex(out.stdout.strip()) == ref.to_bytes() and "12 proofs from 2 host thread(s)" in out.stderr
    # a different public-input count is a different statement (Vandermonde fold, transcript): other bytes
    out = subprocess.run([str(exe), str(cache), "0"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and bytes.fromhex(out.stdout.strip()) != ref.to_bytes()
    # missing files are reported with a status, not a crash
    out = subprocess.run([str(exe), str(tmp_path / "nowhere"), "2"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 1 and ("i/o" in out.stderr.lower() or "io error" in out.stderr.lower()), (out.returncode, out.stderr)


def test_config5_sparse_2_22_through_cache_dir(dvp, tmp_path):
    """BASELINE config #5 at its stated size: an SP1-like sparse R1CS with ~2^22 rows (3.67 M real rows padded to
    2^22, 1..8 terms per side, 16.5 M + 16.5 M + 3.7 M terms) is written in the gnark dump format
    (src/gnark_r1cs.rs:84-91) with its witness file (:58-77,188-198), set up into a cache_dir (SRS point files in the
    reference's format, src/io_utils.rs:83-111) and proved through dvp_prove_cache_dir = Proof::prove(cache_dir, ..)
    (src/proving.rs:426).  Oracle-backed as at 2^20 (tests/fullsize.py): sampled rows by eval_row, sampled domain / SRS
    scalars by their definitions, commit_p and kzg_k by the discrete-log identity, alpha by the oracle transcript, and
    a0 / b0 by the designated-verifier equation on discrete logs."""
    import fullsize as fs
    from util import from_limbs

    A, g = dvp.artifacts, dvp.gnark_r1cs
    log_m = 22
    rnd = random.Random(522)
    inst0, pub_l, prv_l = g.synthetic_sparse_fast(log_m)
    cache = tmp_path / "sp1_like"
    cache.mkdir()
    inst0.write_dump_file(cache / A.R1CS_CONSTRAINTS_FILE)
    w = np.concatenate([dvp.fr.vec([1]), pub_l, prv_l])
    g.write_witness_to_file(cache / A.R1CS_WITNESS_FILE, w)
    trap = (rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    td = dvp.srs.Trapdoor(*trap)
    inst, pv_setup, scalars = dvp.srs.verifier_runs_setup_cache_dir(td, cache, 2, write_precomputes=False, return_scalars=True)
    assert (inst.n_rows, inst.num_constraints, inst.n_wires) == (inst0.n_rows, 1 << log_m, inst0.n_wires)
    assert inst.n_rows > 3_600_000
    D, D2 = fs.check_domains(pv_setup, log_m, rnd, samples=3)
    fs.check_srs_scalars(inst, trap, D, D2, *scalars, rnd, samples=1)
    pv_setup.close()
    del D2
    wf = g.load_witness_from_file(cache / A.R1CS_WITNESS_FILE)
    assert np.array_equal(wf, w)
    pub = from_limbs(wf[1:3])
    proof = dvp.proving.Proof.prove(cache, wf[1:3], wf[3:])
    pv = dvp.proving.Prover.of_cache_dir(cache, 2, inst.num_constraints)
    assert pv.msm_plan(1)[0] >= 18  # the fixed-base tables are in use at this size
    # R1CS evaluation (get_matrix_evaluations_from_witness, src/proving.rs:348-403) on sampled rows, incl. a padded one
    coeffs, a_vec = from_limbs(inst.coeffs), pv.debug("a")
    for i in [0, inst.n_rows - 1] + [rnd.randrange(inst.n_rows) for _ in range(6)]:
        lo, hi = int(inst.l.row_ptr[i]), int(inst.l.row_ptr[i + 1])
        terms = [(int(inst.l.wire[k]), int(inst.l.coeff[k])) for k in range(lo, hi)]
        ww = {t[0]: from_limbs(w[t[0]:t[0] + 1])[0] for t in terms}
        assert from_limbs(a_vec[i:i + 1])[0] == sum(coeffs[c] * ww[a] for a, c in terms) % o.P, i
    assert not a_vec[inst.n_rows:].any()
    fs.check_proof(dvp, pv, inst, trap, pub, None, proof, scalars, check_bary=False, w=w)
    assert dvp.srs.verify(td, pub, proof)
    # an unsatisfied witness is reported with its row, not proved (assert_eq!, src/proving.rs:389-395)
    bad = wf[3:].copy()
    bad[12345, 0] ^= np.uint64(1)
    with pytest.raises(dvp.DvpError) as e:
        dvp.proving.Proof.prove(cache, wf[1:3], bad)
    assert e.value.status == -3 and 0 <= e.value.index < inst.n_rows
    dvp.proving.release_cache_dir()
