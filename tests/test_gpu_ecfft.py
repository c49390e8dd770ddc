"""GPU parity tests (-m gpu): ECFFT extend / enter / exit and the Fr vector kernels through the C ABI,
bit-exact against the oracle at small sizes, and through size-independent identities at the
BASELINE sizes (config #3: 2^20 coefficients; the prover's extend at m = 2^20)."""
import json
import os
import random

import numpy as np
import pytest

import c_oracle as co
import pyref as o
from util import to_limbs, from_limbs, rand_fr_np

pytestmark = pytest.mark.gpu
VEC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))


def H(s):
    return int(s, 16)


@pytest.fixture(scope="module")
def big_tree(dvp):
    return dvp.ec_fft.FFTree(1 << 21)


def test_leaves_and_golden(dvp):
    for key in ("4", "6"):
        g = VEC["ecfft"][key]
        t = dvp.ec_fft.FFTree(1 << int(key))
        assert [hex(x) for x in from_limbs(t.leaves())] == g["leaves"]
        out = t.extend(to_limbs([H(x) for x in g["extend_in"]]))
        assert [hex(x) for x in from_limbs(out)] == g["extend_out"]
        ent = t.enter(to_limbs([H(x) for x in g["coeffs"]]))
        assert [hex(x) for x in from_limbs(ent)] == g["enter_out"]
        assert [hex(x) for x in from_limbs(t.exit(ent))] == g["coeffs"]
    ts = dvp.ec_fft.FFTree(16, True, 5)
    assert [hex(x) for x in from_limbs(ts.leaves())] == VEC["ecfft"]["4_shifted_base5"]["leaves"]


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8])
def test_extend_enter_exit_vs_oracle(dvp, log_n):
    rnd = random.Random(log_n)
    t = dvp.ec_fft.FFTree(1 << log_n)
    ot = o.FFTree(log_n)
    assert from_limbs(t.leaves()) == ot.leaves()
    m = (1 << log_n) // 2
    for batch in (1, 3, 4):
        ev = [rnd.randrange(o.P) for _ in range(batch * m)]
        out = t.extend(to_limbs(ev).reshape(batch, m, 4))
        for b in range(batch):
            assert from_limbs(out[b]) == ot.extend(ev[b * m:(b + 1) * m])
    c = [rnd.randrange(o.P) for _ in range(1 << log_n)]
    e = t.enter(to_limbs(c))
    assert from_limbs(e) == ot.enter(c)
    assert from_limbs(t.exit(e)) == c
    ev = [rnd.randrange(o.P) for _ in range(1 << log_n)]
    assert from_limbs(t.exit(to_limbs(ev))) == ot.exit(ev)


@pytest.mark.parametrize("log_n,shifted", [(12, False), (13, False), (13, True), (14, False)])
def test_multiblock_extend_enter_exit_vs_oracle(dvp, log_n, shifted):
    """The multi-block ECFFT path -- k_butterfly top layers + multi-block k_extend_fused, which run from 2^12 values up
    (FUSE_LOG = 11) -- element for element against the oracle's FFTree (oracle/dvp_oracle_ecfft.c = pyref.FFTree in C,
    pinned against pyref in tests/test_oracle_ecfft.py): leaves, extend (batch 1 and 3), enter, exit of evaluations that
    are NOT an enter image, exit o enter, on the plain and the shifted tree (src/ec_fft.rs:151-155)."""
    n = 1 << log_n
    base = log_n + 1 if shifted else 0
    t = dvp.ec_fft.FFTree(n, shift_by_one=shifted, base_log_n=base)
    ot = co.FFTree(log_n, shifted, base if shifted else None)
    assert np.array_equal(t.leaves(), ot.leaves())
    m = n // 2
    for batch in (1, 3):
        ev = rand_fr_np(batch * m, 10 * log_n + batch).reshape(batch, m, 4)
        out = t.extend(ev)
        for b in range(batch):
            assert np.array_equal(out[b], ot.extend(ev[b])), (batch, b)
    c = rand_fr_np(n, 77 + log_n)
    e = t.enter(c)
    assert np.array_equal(e, ot.enter(c))
    assert np.array_equal(t.exit(e), c)
    ev = rand_fr_np(n, 99 + log_n)
    assert np.array_equal(t.exit(ev), ot.exit(ev))
    t.close()


@pytest.mark.parametrize("log_m", [12, 13, 14, 15, 16, 17, 18, 19, 21])
def test_extend_pass_groupings_agree(dvp, nat, log_m):
    """The unfused top of an extend runs three layers per pass (k_butterfly8), then two (k_butterfly4) or one (k_butterfly) for
    what is left: log_m - 11 = 1..6 top layers covers every grouping ([1], [2], [3], [3,1], [3,2], [3,3]).  Every batch shape
    of the kernels (1..5 vectors) must give the same values with three, two and one layer per pass (DVP_ECFFT_RADIX4 = 2 / 1 /
    0), and the oracle's, element for element, up to m = 2^14; extreme inputs (0, p - 1) ride along: the lazy 30-bit-limb
    values inside an extend (fr.cuh) are only reduced by its last pass.  Round 5: DVP_ECFFT_RADIX4 = 3 (the default) runs 4..9 top
    layers in ONE LDS-tiled launch (k_extend_top): log_m = 15..19 are its tile shapes with 128 .. 8 columns (the prover's own 2^20
    has 4: test_extend_2_20_and_enter_2_16_vs_oracle), log_m = 21 the case with a grouped pass above the tiled ones."""
    m = 1 << log_m
    t = dvp.ec_fft.FFTree(2 * m)
    ot = co.FFTree(log_m + 1) if log_m <= 14 else None
    for batch in ((1, 2, 3, 4, 5) if log_m < 21 else (1, 3)):
        ev = rand_fr_np(batch * m, 31 * log_m + batch).reshape(batch, m, 4)
        ev[0, :7] = to_limbs([0, o.P - 1, 1, o.P - 2, 0, o.P - 1, o.P - 1])
        ev[batch - 1, m // 2:] = to_limbs([o.P - 1] * (m // 2))
        outs = []
        for radix in (3, 2, 1, 0):
            with nat.tune(DVP_ECFFT_RADIX4=radix):
                outs.append(t.extend(ev))
        assert all(np.array_equal(outs[0], x) for x in outs[1:]), batch
        if ot is not None and batch in (1, 3):
            for b in range(batch):
                assert np.array_equal(outs[0][b], ot.extend(ev[b])), (batch, b)
    if ot is not None:
        ot.close()
    t.close()


def test_extend_2_20_and_enter_2_16_vs_oracle(dvp, big_tree):
    """the prover's own extend (m = 2^20, batch 3: a, b, c'; src/proving.rs:410-422) and a 2^16-coefficient enter,
    every element against the oracle's FFTree"""
    m = 1 << 20
    ot = co.FFTree(21)
    assert np.array_equal(big_tree.leaves(), ot.leaves())
    ev = rand_fr_np(3 * m, 2020).reshape(3, m, 4)
    out = big_tree.extend(ev)
    for b in range(3):
        assert np.array_equal(out[b], ot.extend(ev[b])), b
    ot.close()
    t = dvp.ec_fft.FFTree(1 << 16)
    o16 = co.FFTree(16)
    c = rand_fr_np(1 << 16, 1616)
    assert np.array_equal(t.enter(c), o16.enter(c))
    t.close()


def _oracle_sections(ot, log_n):
    """f / recombine_matrices / decompose_matrices of an FFTree in the reference's heap layout (tree_io.py states it),
    assembled from the ORACLE's layers and matrices -- nothing here comes from the product"""
    N = 1 << log_n
    f = np.zeros((2 * N, 4), dtype=np.uint64)
    for d in range(log_n + 1):
        f[N >> d:2 * (N >> d)] = ot.layer(d)
    ident = np.zeros((4, 4), dtype=np.uint64)
    ident[0, 0] = ident[3, 0] = 1
    rec = np.tile(ident, (N, 1)).reshape(N, 4, 4)
    dec = rec.copy()
    n = N // 2
    mats = {(te, w): ot.matrices(bool(te), w).reshape(n - 1, 4, 4) for te in (0, 1) for w in (0, 1)}
    for d in range(log_n - 1):
        nd, off = n >> d, n - (n >> d)
        sl = slice(off, off + nd // 2)
        dec[nd:2 * nd:2], dec[nd + 1:2 * nd:2] = mats[(0, 0)][sl], mats[(1, 0)][sl]
        rec[nd:2 * nd:2], rec[nd + 1:2 * nd:2] = mats[(1, 1)][sl], mats[(0, 1)][sl]
    return {"f": f, "recombine_matrices": rec.reshape(4 * N, 4), "decompose_matrices": dec.reshape(4 * N, 4)}, mats


@pytest.mark.parametrize("log_n", [2, 4, 8, 12])
def test_butterfly_matrices_and_minimal_tree_file(dvp, nat, tmp_path, log_n):
    """SURVEY 8f-4, the half beyond the leaves: the minimal reader of src/tree_io.rs:353-433 loads f, recombine_matrices and
    decompose_matrices.  (1) the 2x2 matrices k_build_mats produced == the oracle's (both directions, every layer);
    (2) a tree file written in the reference's section layout FROM THE ORACLE is accepted by check_tree_file(matrices=True)
    against the regenerated tree, section for section; (3) what write_minimal_tree_file writes from the device tables reads
    back through dvp_fftr_read_fr identical to those sections; a damaged matrix entry is refused with its index."""
    N = 1 << log_n
    t = dvp.ec_fft.FFTree(N)
    ot = co.FFTree(log_n)
    want, mats = _oracle_sections(ot, log_n)
    n = N // 2
    for te in (0, 1):
        for w in (0, 1):
            a = np.zeros(((n - 1) * 4, 4), dtype=np.uint64)
            if n > 1:
                dvp.check(dvp.lib.dvp_debug_ecfft_matrices(t._h, te, w, nat.ptr(a)))
            assert np.array_equal(a.reshape(n - 1, 4, 4), mats[(te, w)]), (te, w)
    path = tmp_path / "tree_oracle"
    dvp.tree_io.write_tree_file(path, want)
    info = dvp.tree_io.check_tree_file(path, t, matrices=True)
    dvp.tree_io.check_tree_file_native(path, t, matrices=True)  # the same comparison inside the library
    assert [s[0] for s in info["sections"]] == ["f", "recombine_matrices", "decompose_matrices"]
    assert [s[1] for s in info["sections"]] == [8 + 29 * 2 * N, 8 + 29 * 4 * N, 8 + 29 * 4 * N]
    mine = tmp_path / "tree_device"
    dvp.tree_io.write_minimal_tree_file(mine, t)
    for name in ("f", "recombine_matrices", "decompose_matrices"):
        assert np.array_equal(dvp.tree_io.read_section(mine, name), want[name]), name
    if log_n >= 4:
        bad = {k: v.copy() for k, v in want.items()}
        entry = 4 * (N // 2 + 3) + 2  # matrix N/2 + 3 (layer 0, pair 3), element m10
        bad["decompose_matrices"][entry, 0] ^= np.uint64(1)
        dvp.tree_io.write_tree_file(path, bad)
        with pytest.raises(ValueError, match=f"decompose_matrices differs.*entry {N // 2 + 3}"):
            dvp.tree_io.check_tree_file(path, t, matrices=True)
        with pytest.raises(ValueError, match=f"decompose_matrices differs.*entry {N // 2 + 3}"):
            dvp.tree_io.check_tree_file_native(path, t, matrices=True)
        dvp.tree_io.check_tree_file_native(path, t, matrices=False)  # the leaves alone are intact
        bad["f"][N + 5, 1] ^= np.uint64(4)  # leaf 5
        dvp.tree_io.write_tree_file(path, bad)
        with pytest.raises(ValueError, match=f"section f differs.*entry {N + 5}"):
            dvp.tree_io.check_tree_file_native(path, t, matrices=False)
    t.close()


def test_edge_vectors(dvp):
    t = dvp.ec_fft.FFTree(64)
    z = np.zeros((32, 4), dtype=np.uint64)
    assert not t.extend(z).any()
    top = to_limbs([o.P - 1] * 32)
    assert from_limbs(t.extend(top)) == [o.P - 1] * 32  # constant polynomial
    bad = to_limbs([o.P] + [0] * 31)  # non-canonical input is rejected, with its index
    with pytest.raises(dvp.DvpError) as ei:
        t.extend(bad)
    assert ei.value.status == -1 and ei.value.index == 0


def test_vanish_at(dvp, nat):
    import ctypes as C
    t = dvp.ec_fft.FFTree(64)
    ot = o.FFTree(6)
    d, d2 = ot.both_domains()
    for which, dom in ((0, d), (1, d2)):
        z = o.poly_from_roots(dom)
        for x in (5, dom[3], o.P - 2):
            out = np.zeros(4, dtype=np.uint64)
            dvp.check(dvp.lib.dvp_ecfft_vanish_at(t._h, which, nat.ptr(to_limbs([x])), nat.ptr(out)))
            assert from_limbs(out[None])[0] == o.poly_eval(z, x)


def test_batch_inverse_and_barycentric(dvp, nat):
    rnd = random.Random(9)
    vals = [rnd.randrange(o.P) for _ in range(1000)]
    vals[7] = 0
    vals[999] = 0
    a = to_limbs(vals)
    dvp.check(dvp.lib.dvp_fr_batch_inverse(nat.ptr(a), len(vals)))
    assert from_limbs(a) == o.fr_batch_inverse(vals)
    # every workgroup shape of k_batch_inverse (DVP_FR_BI_SHAPE) on a length that leaves ragged last blocks, zeros included
    big = [rnd.randrange(o.P) for _ in range(20011)]
    for z in (0, 63, 64, 4095, 4096, 8191, 8192, 20010):
        big[z] = 0
    exp = o.fr_batch_inverse(big)
    for shape in (0, 1, 2, 3, 4, 5, 6, 7):
        with nat.tune(DVP_FR_BI_SHAPE=shape):
            b = to_limbs(big)
            dvp.check(dvp.lib.dvp_fr_batch_inverse(nat.ptr(b), len(big)))
            assert from_limbs(b) == exp, shape
    ot = o.FFTree(6)
    tb = o.domain_tables(ot)
    ev = [rnd.randrange(o.P) for _ in range(tb["m"])]
    alpha = rnd.randrange(o.P)
    z_alpha = o.poly_eval(tb["z_poly"], alpha)
    out = np.zeros(4, dtype=np.uint64)
    dvp.check(dvp.lib.dvp_barycentric_eval(nat.ptr(to_limbs(tb["D"])), nat.ptr(to_limbs(tb["bar_wts"])), nat.ptr(to_limbs([z_alpha])),
                                           nat.ptr(to_limbs(ev)), tb["m"], nat.ptr(to_limbs([alpha])), nat.ptr(out)))
    exp = o.barycentric_eval(tb["D"], tb["bar_wts"], z_alpha, ev, alpha)
    assert from_limbs(out[None])[0] == exp == o.lagrange_eval(tb["D"], ev, alpha)


def test_extend_full_size_properties(dvp, big_tree):
    """m = 2^20 (the prover's extend, src/proving.rs:410-422): degree-1 polynomial, linearity."""
    m = 1 << 20
    leaves = big_tree.leaves()
    d, d2 = from_limbs(leaves[0:64:2]), from_limbs(leaves[1:64:2])
    all_d = leaves[0::2]
    # P(x) = a + b x on D  ->  must come out as a + b x on D'
    a, b = 1234567, 7654321
    # evaluate a + b*d_i with numpy object ints only on a sample; build the full vector on the GPU side by
    # linearity: extend(const a) + b * extend(d)  ==  extend(a + b d)
    x = np.ascontiguousarray(all_d)
    ext_d = big_tree.extend(x)  # extend of the identity polynomial P(x) = x
    assert from_limbs(ext_d[:32]) == d2
    assert from_limbs(ext_d[-3:]) == from_limbs(leaves[-5::2])
    r1, r2 = rand_fr_np(m, 1), rand_fr_np(m, 2)
    s = to_limbs([(p + q) % o.P for p, q in zip(from_limbs(r1[:4096]), from_limbs(r2[:4096]))])
    both = big_tree.extend(np.stack([r1, r2]))
    # linearity on a prefix is not meaningful (extend is global), so check it on full vectors of a smaller tree
    t = dvp.ec_fft.FFTree(1 << 13)
    e = t.extend(np.stack([r1[:4096], r2[:4096], s]))
    assert from_limbs(e[2]) == [(p + q) % o.P for p, q in zip(from_limbs(e[0]), from_limbs(e[1]))]
    assert both.shape == (2, m, 4)


def test_enter_exit_roundtrip_2_20(dvp):
    """BASELINE config #3: 2^20 coefficients, enter then exit, exact round trip; monomial spot checks."""
    n = 1 << 20
    t = dvp.ec_fft.FFTree(n)
    c = rand_fr_np(n, 3)
    e = t.enter(c)
    back = t.exit(e)
    assert (back == c).all()
    # enter of a X^j + b X^k is a L^j + b L^k on every leaf: check 64 random leaves
    rnd = random.Random(4)
    j, k, a, b = rnd.randrange(n), rnd.randrange(n), rnd.randrange(o.P), rnd.randrange(o.P)
    sp = np.zeros((n, 4), dtype=np.uint64)
    sp[j] = to_limbs([a])[0]
    sp[k] = to_limbs([b])[0]
    ev = t.enter(sp)
    leaves = t.leaves()
    for idx in [0, 1, n - 1] + [rnd.randrange(n) for _ in range(61)]:
        L = from_limbs(leaves[idx][None])[0]
        assert from_limbs(ev[idx][None])[0] == (a * pow(L, j, o.P) + b * pow(L, k, o.P)) % o.P


@pytest.mark.parametrize("log_n", [1, 2, 3, 6, 10, 11, 12, 13, 15])
def test_enter_exit_folded_stages_vs_oracle_and_unfolded(dvp, nat, log_n):
    """Round 6: enter / exit fold their pointwise stages (k_exit_pre / _mid / _mulc / _post, the h1 -> h0 copy, k_enter_combine for
    sub-vectors up to 1024) into the first / last pass of their extends (ExtIo in ecfft.hip).  Every level shape is crossed here:
    h = 1 (never folded), single-launch extends (h <= 2048), and extends with k_extend_top in front and behind (h >= 4096, incl. the
    short-vector tile walk); folded and unfolded (DVP_ECFFT_FOLD=0) must both equal the oracle element for element, on enter
    images and on evaluations that are no enter image."""
    n = 1 << log_n
    t = dvp.ec_fft.FFTree(n)
    ot = co.FFTree(log_n)
    c = rand_fr_np(n, 600 + log_n)
    ev = rand_fr_np(n, 700 + log_n)
    want_e, want_x = ot.enter(c), ot.exit(ev)
    for fold in (1, 0, 1):
        with nat.tune(DVP_ECFFT_FOLD=fold):
            e = t.enter(c)
            assert np.array_equal(e, want_e), fold
            assert np.array_equal(t.exit(e), c), fold
            assert np.array_equal(t.exit(ev), want_x), fold
    ot.close()
    t.close()


def test_exit_enter_2_20_folded_equals_unfolded(dvp, nat):
    """the same at BASELINE config #3's size, where the oracle takes minutes: both flavours give the same 2^20 values (the unfolded one
    is what rounds 1-5 checked against the oracle up to 2^16 and by the exact round trip here)"""
    import torch
    n = 1 << 20
    t = dvp.ec_fft.FFTree(n)
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(rand_fr_np(n, 2026).view(np.int64)).cuda()
    outs = {}
    for fold in (0, 1):
        with nat.tune(DVP_ECFFT_FOLD=fold):
            d_x, d_e = torch.empty_like(d_in), torch.empty_like(d_in)
            t.exit_dev(d_in.data_ptr(), d_x.data_ptr(), st)
            t.enter_dev(d_in.data_ptr(), d_e.data_ptr(), st)
            torch.cuda.synchronize()
            outs[fold] = (d_x, d_e)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # in place (d_out == d_in) is allowed by both entries
    d_c = d_in.clone()
    t.enter_dev(d_c.data_ptr(), d_c.data_ptr(), st)
    assert torch.equal(d_c, outs[1][1])
    t.exit_dev(d_c.data_ptr(), d_c.data_ptr(), st)
    assert torch.equal(d_c, d_in)
    t.close()


def test_extend_consistent_with_enter(dvp):
    """evaluations of one polynomial (degree < m) on D and D' via enter on the 2m tree must be linked by extend."""
    t = dvp.ec_fft.FFTree(1 << 12)
    m = 1 << 11
    c = np.zeros((2 * m, 4), dtype=np.uint64)
    c[:m] = rand_fr_np(m, 5)
    e = t.enter(c)
    assert (t.extend(np.ascontiguousarray(e[0::2])) == e[1::2]).all()


def test_tree_file_checked_against_regenerated_domain(dvp, tmp_path):
    """an FFTR tree file (src/tree_io.rs) whose leaves are the oracle's tree2n leaves (what a reference-built
    file holds: leaves x(C + i*G'), src/ec_fft.rs:158-162) is accepted by check_tree_file against the device-regenerated
    domain, for the plain and the shifted tree; a file with one damaged leaf is refused with the leaf index."""
    log_n = 7
    n = 1 << log_n
    for shifted in (False, True):
        ot = o.FFTree(log_n, shifted=shifted)
        leaves = ot.leaves()
        path = tmp_path / f"tree_{int(shifted)}"
        dvp.tree_io.write_tree_file(path, {"f": to_limbs([0] * n + leaves)})
        t = dvp.ec_fft.FFTree(n, shift_by_one=shifted, base_log_n=log_n if shifted else 0)
        info = dvp.tree_io.check_tree_file(path, t)
        assert info["leaves"] == n and info["sections"][0][0] == "f"
        bad = list(leaves)
        bad[37] = (bad[37] + 1) % o.P
        dvp.tree_io.write_tree_file(path, {"f": to_limbs([0] * n + bad)})
        with pytest.raises(ValueError, match="leaf 37"):
            dvp.tree_io.check_tree_file(path, t)
        t.close()
