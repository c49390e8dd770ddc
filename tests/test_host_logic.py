"""CPU tests of the host-side mirror (no GPU compute): R1CS containers and file formats, CSR transpose,
Proof bit packing, shard ranges.  These import the product package (the C-ABI library must load) but
never call a kernel."""
import random

import numpy as np

import pyref as o


def test_r1cs_dump_roundtrip_and_rows(dvp):
    g = dvp.gnark_r1cs
    inst = g.R1CSInstance.from_rows(g.TOY_ROWS, g.TOY_COEFFS, 2)
    assert (inst.num_constraints, inst.n_rows, inst.n_wires) == (8, 5, 8)
    blob = inst.to_dump_bytes()
    # layout of src/gnark_r1cs.rs:84-91: u32 nbCoeffs, 32-byte BE coefficients, u32 nbRows, per row 3 counts + terms
    assert blob[:4] == (2).to_bytes(4, "little") and blob[4:36] == (1).to_bytes(32, "big")
    back = g.R1CSInstance.from_dump_bytes(blob, 2)
    for a, b in ((inst.l, back.l), (inst.r, back.r), (inst.o, back.o)):
        assert (a.row_ptr == b.row_ptr).all() and (a.wire == b.wire).all() and (a.coeff == b.coeff).all()
    assert (inst.coeffs == back.coeffs).all()


def test_csr_transpose(dvp):
    g = dvp.gnark_r1cs
    inst, pub, prv = g.synthetic_sparse(6)
    t = inst.l.transpose(inst.n_wires)
    # every (row, wire, coeff) triple appears exactly once on both sides
    fwd = sorted((r, int(inst.l.wire[k]), int(inst.l.coeff[k])) for r in range(inst.n_rows)
                 for k in range(inst.l.row_ptr[r], inst.l.row_ptr[r + 1]))
    bwd = sorted((int(t.wire[k]), w, int(t.coeff[k])) for w in range(inst.n_wires) for k in range(t.row_ptr[w], t.row_ptr[w + 1]))
    assert fwd == bwd
    # the synthetic witness satisfies every row (src/proving.rs:389-395)
    w = [1] + pub + prv
    coeffs = dvp.fr.to_ints(inst.coeffs)
    for r in range(inst.n_rows):
        def ev(m):
            return sum(coeffs[int(m.coeff[k])] * w[int(m.wire[k])] for k in range(m.row_ptr[r], m.row_ptr[r + 1])) % o.P
        assert ev(inst.l) * ev(inst.r) % o.P == ev(inst.o)


def test_synthetic_dense_witness_satisfies(dvp):
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(7)
    w = [1] + pub + prv
    coeffs = dvp.fr.to_ints(inst.coeffs)
    assert inst.n_rows == inst.num_constraints == inst.n_wires == 128
    for r in range(inst.n_rows):
        def ev(m):
            return sum(coeffs[int(m.coeff[k])] * w[int(m.wire[k])] for k in range(m.row_ptr[r], m.row_ptr[r + 1])) % o.P
        assert ev(inst.l) * ev(inst.r) % o.P == ev(inst.o), r


def test_proof_bits_match_reference_layout(dvp):
    rnd = random.Random(1)
    pr = dvp.proving.Proof(bytes(rnd.randrange(256) for _ in range(30)), bytes(rnd.randrange(256) for _ in range(30)),
                           (12345).to_bytes(29, "little"), (o.P - 1).to_bytes(29, "little"))
    bits = pr.to_bits()
    assert len(bits) == 944 and bits == o.proof_to_bits(pr.commit_p, pr.kzg_k, 12345, o.P - 1)
    assert dvp.proving.Proof.from_bits(bits) == pr
    assert pr.a0_fr() == (12345, True) and pr.b0_fr() == (o.P - 1, True)
    bad = dvp.proving.Proof(pr.commit_p, pr.kzg_k, o.P.to_bytes(29, "little"), pr.b0)
    assert bad.a0_fr()[1] is False  # FrBits::to_fr rejects values >= p, src/curve.rs:55-57


def test_fr_limb_helpers(dvp):
    vals = [0, 1, o.P - 1, 1 << 200]
    arr = dvp.fr.vec(vals)
    assert arr.shape == (4, 4) and dvp.fr.to_ints(arr) == vals
    assert dvp.fr.to_int(dvp.fr.limbs(o.P + 5)) == 5


def test_shard_plan_roles_and_coverage(dvp):
    """distributed.shard_plan: contiguous cover of both MSM index spaces, ranks that skip the extends stay inside the
    extend-free parts, and the modelled loads are balanced within a few percent"""
    d = dvp.distributed
    for n_wires, m in ((1 << 20, 1 << 20), (3_000_000, 1 << 22), (8, 8), (100, 1 << 10)):
        for world in (1, 2, 3, 4, 5, 6, 7, 8):
            plan = d.shard_plan(world, n_wires, m)
            assert len(plan) == world
            for idx, total in ((0, n_wires + m), (1, 4 * m)):
                rs = [p[idx] for p in plan]
                assert rs[0][0] == 0 and rs[-1][1] == total
                assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
                assert all(lo <= hi for lo, hi in rs)
            ext = [p[2] for p in plan]
            assert ext[-1] and ext == sorted(ext)  # extenders are the last ranks, at least one
            for (a, b, need) in plan:
                if not need:
                    assert a[1] <= n_wires and b[1] <= 2 * m
            if m >= 1 << 20:
                e = (0.42 if sum(ext) < 3 else 0.30) * m  # the model's cost of the extends (by vector from three extenders up)
                loads = [(a[1] - a[0]) + (b[1] - b[0]) + (e if need else 0) for a, b, need in plan]
                assert max(loads) <= 1.03 * (sum(loads) / world) or all(ext)
    assert all(need for _, _, need in d.shard_plan(2, 1 << 20, 1 << 20))       # small worlds: uniform
    assert not d.shard_plan(8, 1 << 20, 1 << 20)[0][2]                           # 8 ranks: some skip the extends
    # MEASURED costs (distributed.measure_plan_costs: {"replicated", "split"} in (scalar, base) pairs) replace the typed-in defaults: the
    # defaults as a dict give the default plan; a free extend makes every rank an extender; a dear exchange shrinks the extender group
    m = 1 << 20
    assert d.shard_plan(8, m, m, extend_pairs={"replicated": 0.42 * m, "split": 0.30 * m}) == d.shard_plan(8, m, m)
    assert all(need for _, _, need in d.shard_plan(8, m, m, extend_pairs={"replicated": 0.0, "split": 0.0}))
    n_default = sum(1 for p in d.shard_plan(8, m, m) if p[2])
    n_dear = sum(1 for p in d.shard_plan(8, m, m, extend_pairs={"replicated": 0.42 * m, "split": 3.0 * m}) if p[2])
    assert 1 <= n_dear < n_default


def test_pin_xsk233_tool_agrees_with_the_oracle_rule():
    """tools/pin_xsk233.py (self-contained arithmetic) must name the CURRENT rule for a vector produced by the oracle's
    candidate codec, and must say so when a vector does not match"""
    import importlib.util
    import os

    import c_oracle as co

    spec = importlib.util.spec_from_file_location("pin_xsk233", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pin_xsk233.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    k = 0xC0FFEE1234567
    vec = co.xsk233_encode(co.k233_mulgen(k))
    lines = []
    assert pin.pin(k, vec, out=lines.append) == 0 and lines[0].startswith("CURRENT RULE CONFIRMED")
    bad = bytes([vec[0] ^ 4]) + vec[1:]
    assert pin.pin(k, bad, out=lines.append) == 2


def test_gf233_word_by_word_reduction_formulas():
    """gf233.cuh (round 5): gf_reduce16 and the Karatsuba combine of the hot multiplier fold the high words with ONE funnel shift per
    fold instead of a word-serial loop.  The formulas, restated on Python ints word for word, against plain polynomial reduction
    modulo z^233 + z^74 + 1: random and all-ones inputs at the full 512 / 466 bits, and (L, M, H) triples at their largest sizes
    (233, 233, 231 bits) -- the bounds that let the combine drop the words 11 and 15."""
    import random
    M32 = 0xFFFFFFFF

    def alignbit(hi, lo, s):
        return (((hi << 32) | lo) >> s) & M32

    def reduce_poly(x):
        while x.bit_length() > 233:
            b = x.bit_length() - 1
            x ^= (1 << b) | (1 << (b - 233 + 74)) | (1 << (b - 233))
        return x

    def words(x, n):
        return [(x >> (32 * i)) & M32 for i in range(n)]

    def tail(r):
        t = r[7] >> 9
        r[0] ^= t
        r[2] ^= (t << 10) & M32
        r[3] ^= t >> 22
        r[7] &= 0x1FF
        return sum(w << (32 * i) for i, w in enumerate(r))

    def reduce16(c):
        T15, T14, T13, T12 = c[15], c[14], c[13], c[12]
        T11 = c[11] ^ (T15 >> 31)
        T10 = c[10] ^ alignbit(T15, T14, 31)
        T9 = c[9] ^ alignbit(T14, T13, 31)
        T8 = c[8] ^ alignbit(T13, T12, 31) ^ (T15 >> 9)
        r = [c[0] ^ ((T8 << 23) & M32), c[1] ^ alignbit(T9, T8, 9), c[2] ^ alignbit(T10, T9, 9),
             c[3] ^ alignbit(T11, T10, 9) ^ ((T8 << 1) & M32), c[4] ^ alignbit(T12, T11, 9) ^ alignbit(T9, T8, 31),
             c[5] ^ alignbit(T13, T12, 9) ^ alignbit(T10, T9, 31), c[6] ^ alignbit(T14, T13, 9) ^ alignbit(T11, T10, 31),
             c[7] ^ alignbit(T15, T14, 9) ^ alignbit(T12, T11, 31)]
        return tail(r)

    def combine(L, H, M):
        M = [M[i] ^ L[i] ^ H[i] for i in range(8)]
        ms = [(M[0] << 21) & M32] + [alignbit(M[i], M[i - 1], 11) for i in range(1, 8)]
        hs = [(H[0] << 10) & M32] + [alignbit(H[i], H[i - 1], 22) for i in range(1, 8)]
        T14, T13, T12, T11 = hs[7], hs[6], hs[5], hs[4]
        T10 = ms[7] ^ hs[3] ^ (T14 >> 31)
        T9 = ms[6] ^ hs[2] ^ alignbit(T14, T13, 31)
        T8 = ms[5] ^ hs[1] ^ alignbit(T13, T12, 31)
        r = [L[0] ^ ((T8 << 23) & M32), L[1] ^ alignbit(T9, T8, 9), L[2] ^ alignbit(T10, T9, 9),
             L[3] ^ ms[0] ^ alignbit(T11, T10, 9) ^ ((T8 << 1) & M32), L[4] ^ ms[1] ^ alignbit(T12, T11, 9) ^ alignbit(T9, T8, 31),
             L[5] ^ ms[2] ^ alignbit(T13, T12, 9) ^ alignbit(T10, T9, 31), L[6] ^ ms[3] ^ alignbit(T14, T13, 9) ^ alignbit(T11, T10, 31),
             L[7] ^ ms[4] ^ hs[0] ^ (T14 >> 9) ^ alignbit(T12, T11, 31)]
        return tail(r)

    rnd = random.Random(2025)
    for it in range(4000):
        bits = (512, 466, 465)[it % 3]
        x = (1 << bits) - 1 if it < 3 else rnd.getrandbits(bits)
        assert reduce16(words(x, 16)) == reduce_poly(x), it
        L, Mm, H = rnd.getrandbits(233), rnd.getrandbits(233), rnd.getrandbits(231)
        if it % 4 == 0:
            L, Mm, H = L | (1 << 232), Mm | (1 << 232), H | (1 << 230)
        if it == 1:
            L, Mm, H = (1 << 233) - 1, (1 << 233) - 1, (1 << 231) - 1
        assert combine(words(L, 8), words(H, 8), words(Mm, 8)) == reduce_poly(L ^ ((Mm ^ L ^ H) << 117) ^ (H << 234)), it
    # the accumulator of a half product is shifted only as far as it is filled (gf_k_mul_tab): the bit lengths the comments claim
    ln = 64 + 119  # the three-word rows k = 10 .. 7: an entry of 119 bits two words up
    for k in (9, 8, 7):
        ln += 3
        assert ln <= 6 * 32, k  # shifted as six words
        ln = max(ln, 64 + 119)
    for k in (6, 5, 4, 3):
        ln += 3
        assert ln <= 7 * 32, k  # seven words
        ln = max(ln, 96 + 119)
    for k in (2, 1, 0):
        ln += 3
        ln = max(ln, 96 + 119)
    assert ln == 233
