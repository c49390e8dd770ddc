"""GPU parity tests (-m gpu): sect233k1 MSM / fixed-base mulgen / codec through the C ABI.
Bit-exact (affine group elements) against the OpenSSL-pinned C oracle; BASELINE config #2
(2^16 points) is checked through the discrete-log identity sum s_i (k_i G) = (sum s_i k_i) G,
exactly the shape of the reference's own test_msm (src/curve.rs:218-232)."""
import json
import os
import random

import numpy as np
import pytest

import pyref as o
import c_oracle as co
from util import to_limbs, from_limbs, pts_to_np, np_to_pt, rand_fr_np, np_dot_mod, np_dot_mod_fast, tau_adversarial_scalars

pytestmark = pytest.mark.gpu
OSSL = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k233_openssl.json")))["vectors"]


def H(s):
    return int(s, 16)


def gpu_msm(dvp, scalars, bases, inf=None):
    xy, is_inf = dvp.curve.multi_scalar_mul(scalars, bases, inf)
    return np_to_pt(xy, is_inf)


def test_mulgen_against_openssl(dvp):
    ks = [H(e["k"]) for e in OSSL]
    xy, inf = dvp.curve.point_scalar_mul_gen_batch(to_limbs(ks + [0]))
    for i, e in enumerate(OSSL):
        assert np_to_pt(xy[i], inf[i]) == (H(e["x"]), H(e["y"]))
    assert inf[-1] == 1  # 0 * G = neutral


def test_msm_edge_cases(dvp):
    g = [co.k233_mulgen(k) for k in (1, 2, 3, 4, 5)]
    G = pts_to_np(g)
    assert gpu_msm(dvp, to_limbs([1]), G[:1]) == g[0]
    assert gpu_msm(dvp, to_limbs([o.P - 1]), G[2:3]) == o.k233_neg(g[2])
    assert gpu_msm(dvp, to_limbs([0, 0, 0]), G[:3]) is None            # all-zero scalars
    assert gpu_msm(dvp, to_limbs([1, o.P - 1]), G[[3, 3]]) is None     # P + (-P)
    assert gpu_msm(dvp, to_limbs([1, 1]), G[[3, 3]]) == co.k233_mulgen(8)  # P + P inside one bucket
    assert gpu_msm(dvp, np.zeros((0, 4), np.uint64), np.zeros((0, 8), np.uint64)) is None  # empty
    inf = np.array([0, 1, 0], dtype=np.uint8)                              # neutral among the bases
    assert gpu_msm(dvp, to_limbs([7, 9, 11]), G[:3], inf) == co.k233_mulgen(7 * 1 + 11 * 3)
    with pytest.raises(dvp.DvpError) as ei:                                 # scalar >= p is rejected
        gpu_msm(dvp, to_limbs([5, o.P]), G[:2])
    assert ei.value.status == -1 and ei.value.index == 1


def test_msm_all_equal_bases_like_reference(dvp):
    """test_msm, src/curve.rs:218-232 (10 000 random scalars on G)."""
    n = 10000
    s = rand_fr_np(n, 11)
    G = np.tile(pts_to_np([o.G_STD]), (n, 1))
    assert gpu_msm(dvp, s, G) == co.k233_mulgen(sum(from_limbs(s)) % o.P)


def test_msm_vs_reference_shaped_cpu(dvp):
    n = 777
    k, s = rand_fr_np(n, 21), rand_fr_np(n, 22)
    bases, inf = dvp.curve.point_scalar_mul_gen_batch(k)
    assert not inf.any()
    assert gpu_msm(dvp, s, bases) == co.msm(s, bases, threads=4)


def test_msm_skewed_scalars(dvp):
    """many equal / tiny scalars: one bucket receives most points (load-balance path of the reduction)."""
    n = 5000
    k = rand_fr_np(n, 31)
    bases, _ = dvp.curve.point_scalar_mul_gen_batch(k)
    s = np.zeros((n, 4), dtype=np.uint64)
    s[:, 0] = 1
    s[::7, 0] = 2
    s[::11, 0] = 0
    assert gpu_msm(dvp, s, bases) == co.k233_mulgen(np_dot_mod(s, k))


@pytest.mark.parametrize("log_n", [12, 16])
def test_msm_config2(dvp, log_n):
    """BASELINE config #2: 2^16 random scalars/bases, bit-exact vs the CPU restatement."""
    n = 1 << log_n
    k, s = rand_fr_np(n, 41 + log_n), rand_fr_np(n, 42 + log_n)
    bases, inf = dvp.curve.point_scalar_mul_gen_batch(k)
    assert not inf.any()
    got = gpu_msm(dvp, s, bases)
    assert got == co.k233_mulgen(np_dot_mod(s, k))
    if log_n == 12:  # and directly against the per-point scalar-mult + add-tree CPU shape
        assert got == co.msm(s, bases, threads=8)


def test_codec_roundtrip_and_invalid(dvp):
    ks = [1, 2, 3, 0xDEADBEEF, o.P - 1] + [random.Random(5).randrange(o.P) for _ in range(20)]
    enc = dvp.curve.point_scalar_mul_gen_batch_bytes(to_limbs(ks + [0]))
    for i, k in enumerate(ks):
        assert enc[i].tobytes() == co.xsk233_encode(co.k233_mulgen(k))
    assert enc[-1].tobytes() == bytes(30)
    xy, inf = dvp.curve.from_bytes(enc)
    for i, k in enumerate(ks):
        assert np_to_pt(xy[i], inf[i]) == co.k233_mulgen(k)
    assert inf[-1] == 1
    assert (dvp.curve.to_bytes(xy, inf) == enc).all()
    rnd = random.Random(6)
    seen_bad = 0
    for _ in range(24):
        w = rnd.getrandbits(233).to_bytes(30, "little")
        exp, ok = co.xsk233_decode(w)
        try:
            x, i_ = dvp.curve.from_bytes(np.frombuffer(w, dtype=np.uint8))
            assert ok and np_to_pt(x[0], i_[0]) == exp
        except dvp.DvpError as e:
            assert not ok and e.status == -2 and e.index == 0
            seen_bad += 1
    assert seen_bad > 0


@pytest.mark.parametrize("rule", range(1, 12))
def test_codec_rule_variants(dvp, rule):
    """dvp_codec_set_rule: every presentation of the encoded element that tools/pin_xsk233.py can name (w / w^2 / sqrt w,
    "+1", byte order) -- GPU encode == the oracle's own formula for that rule, GPU decode inverts it, invalid bytes agree, and a
    proof-shaped round trip (mulgen -> bytes -> wire-format MSM) holds under the rule.  Rule 0 is the default everywhere else."""
    ks = [1, 2, 3, o.P - 1] + [random.Random(50 + rule).randrange(o.P) for _ in range(12)]
    try:
        dvp.check(dvp.lib.dvp_codec_set_rule(rule))
        assert dvp.lib.dvp_codec_get_rule() == rule
        enc = dvp.curve.point_scalar_mul_gen_batch_bytes(to_limbs(ks + [0]))
        for i, k in enumerate(ks):
            assert enc[i].tobytes() == o.xsk233_encode(co.k233_mulgen(k), rule), (rule, k)
            assert o.xsk233_decode(enc[i].tobytes(), rule) == (co.k233_mulgen(k), True)
        assert enc[-1].tobytes() == bytes(30)
        xy, inf = dvp.curve.from_bytes(enc)
        for i, k in enumerate(ks):
            assert np_to_pt(xy[i], inf[i]) == co.k233_mulgen(k)
        assert inf[-1] == 1 and (dvp.curve.to_bytes(xy, inf) == enc).all()
        rnd = random.Random(60 + rule)
        for _ in range(8):
            w = rnd.getrandbits(233).to_bytes(30, "big" if rule & 2 else "little")
            exp, ok = o.xsk233_decode(w, rule)
            try:
                x, i_ = dvp.curve.from_bytes(np.frombuffer(w, dtype=np.uint8))
                assert ok and np_to_pt(x[0], i_[0]) == exp
            except dvp.DvpError as e:
                assert not ok and e.status == -2
        n = 64
        k, s = rand_fr_np(n, 70 + rule), rand_fr_np(n, 71 + rule)
        out = dvp.curve.multi_scalar_mul_bytes(s.view(np.uint8).reshape(n, 32), dvp.curve.point_scalar_mul_gen_batch_bytes(k))
        assert out == o.xsk233_encode(co.k233_mulgen(np_dot_mod(s, k)), rule)
    finally:
        dvp.check(dvp.lib.dvp_codec_set_rule(0))
    assert dvp.lib.dvp_codec_set_rule(12) == -1 and dvp.lib.dvp_codec_get_rule() == 0


def test_msm_wire_format(dvp):
    """scalars 32-B LE + bases 30-B encodings in, 30-B encoding out (the reference's file payloads)."""
    n = 300
    k, s = rand_fr_np(n, 51), rand_fr_np(n, 52)
    enc = dvp.curve.point_scalar_mul_gen_batch_bytes(k)
    out = dvp.curve.multi_scalar_mul_bytes(s.view(np.uint8).reshape(n, 32), enc)
    assert out == co.xsk233_encode(co.k233_mulgen(np_dot_mod(s, k)))


@pytest.mark.parametrize("slide", [0, 3])  # signed aligned binary windows (the default) / tau-adic aligned windows
@pytest.mark.parametrize("hint", [0, 1 << 10, 1 << 24])
def test_fixed_base_msm_context(dvp, hint, slide):
    """dvp_msm_ctx_*: pre-rotated bases, shared bucket set; full range, sub-ranges (the per-GPU shards) and a
    neutral base, for several window sizes (range_hint drives the choice; 2^24 forces c = 20, two-level sort), with both
    table flavours (signed binary windows over rows 2^(o_w) P, tau-adic windows over rows tau^(o_w) P)."""
    n = 6000
    k, s = rand_fr_np(n, 61), rand_fr_np(n, 62)
    bases, _ = dvp.curve.point_scalar_mul_gen_batch(k)
    inf = np.zeros(n, dtype=np.uint8)
    inf[5] = 1
    knobs = {"DVP_MSM_ALIGNED_SIGNED": 0 if slide == 3 else 1}
    if hint <= n:
        with dvp.tune(**knobs):
            fb = dvp.curve.FixedBaseMsm(bases, inf, hint)
    else:
        with dvp.tune(DVP_MSM_FIXED_C=20, **knobs):
            fb = dvp.curve.FixedBaseMsm(bases, inf, 0)
    ks, ss = from_limbs(k), from_limbs(s)

    def expect(lo, hi):
        return co.k233_mulgen(sum(ss[i] * ks[i] for i in range(lo, hi) if i != 5) % o.P)

    xy, is_inf = fb.run(s)
    assert np_to_pt(xy, is_inf) == expect(0, n)
    # pair rounds all the way down (normally only while a round has >= 2^19 additions), with 1, 3 and 64 additions per
    # shared inversion at most (the device picks the count per round; one chip-full holds this whole input)
    for bmax in (1, 3, 64):
        with dvp.tune(DVP_MSM_AFF_MIN=64, DVP_MSM_AFF_BMAX=bmax):
            xy, is_inf = fb.run(s)
            assert np_to_pt(xy, is_inf) == expect(0, n)
            xy, is_inf = fb.run(s[7:4000], 7, 4000)
            assert np_to_pt(xy, is_inf) == expect(7, 4000)
    parts = []
    for lo, hi in ((0, 1500), (1500, 1501), (1501, 6000), (10, 10)):
        xy, is_inf = fb.run(s[lo:hi], lo, hi)
        got = np_to_pt(xy, is_inf)
        assert got == (expect(lo, hi) if hi > lo else None)
        parts.append(got)
    acc = None
    for pt in parts:
        acc = o.k233_add(acc, pt)
    assert acc == expect(0, n)  # the shards add up to the whole (what the all-gather + local add relies on)
    # a scalar >= p is refused with its index, whichever kernel recodes (level 1 of the sort reads the scalars itself by default)
    bad = s.copy()
    bad[4321] = to_limbs([o.P])[0]
    for fused in (0, 1):
        with dvp.tune(DVP_MSM_SORT_FUSED=fused):
            with pytest.raises(dvp.DvpError) as ei:
                fb.run(bad)
            assert ei.value.status == -1 and ei.value.index == 4321
            xy, is_inf = fb.run(s)
            assert np_to_pt(xy, is_inf) == expect(0, n)


def test_msm_heavy_skew_2_18(dvp):
    """2^18 scalars drawn from {0, 1, 2, 3}: a handful of buckets receive ~2^16 points each (R1CS witnesses
    look like this: mostly bits and small values).  Exercises ~17 pair rounds and the skew-balanced reducer."""
    n = 1 << 18
    k = rand_fr_np(n, 71)
    bases, _ = dvp.curve.point_scalar_mul_gen_batch(k)
    rng = np.random.default_rng(72)
    s = np.zeros((n, 4), dtype=np.uint64)
    s[:, 0] = rng.integers(0, 4, size=n)
    assert gpu_msm(dvp, s, bases) == co.k233_mulgen(np_dot_mod(s, k))
    fb = dvp.curve.FixedBaseMsm(bases)
    xy, is_inf = fb.run(s)
    assert np_to_pt(xy, is_inf) == co.k233_mulgen(np_dot_mod(s, k))


def test_bucket_reduction_exceptional_pairs(dvp):
    """what the pair rounds leave (k_bucket_pairs / k_bucket_rest): bases drawn from eight points and their negatives,
    scalars from five values, so that the buckets the rounds hand over hold equal points (P + P), opposite points
    (P - P) and the infinity markers earlier rounds made of them -- the pairs the fast formulas refuse and the list
    redoes.  Fixed-base and one-shot, the rounds stopped at several depths, against the fan-in-K reducer route
    (DVP_MSM_BUCKET_PAIRS_MAX=0) and the oracle."""
    rnd = random.Random(777)
    n = 4096
    k8 = [rnd.randrange(1, o.P) for _ in range(8)]
    p8 = [co.k233_mulgen(x) for x in k8]
    vals = [rnd.randrange(o.P) for _ in range(4)] + [1]
    pts, ks, sv = [], [], []
    for i in range(n):
        j, neg = rnd.randrange(8), rnd.random() < 0.5
        pts.append(o.k233_neg(p8[j]) if neg else p8[j])
        ks.append(o.P - k8[j] if neg else k8[j])
        sv.append(rnd.choice(vals))
    bases, s = pts_to_np(pts), to_limbs(sv)
    exp = co.k233_mulgen(sum(a * b for a, b in zip(sv, ks)) % o.P)
    lo, hi = 100, 3000
    exp_part = co.k233_mulgen(sum(a * b for a, b in zip(sv[lo:hi], ks[lo:hi])) % o.P)
    for c in (8, 11):
        with dvp.tune(DVP_MSM_FIXED_C=c):
            fb = dvp.curve.FixedBaseMsm(bases)
        for aff_min in (32, 512, 4096):
            for pairs_max in (0, 12, 100000):
                with dvp.tune(DVP_MSM_AFF_MIN=aff_min, DVP_MSM_BUCKET_PAIRS_MAX=pairs_max, DVP_MSM_C=c, DVP_MSM_ROUND_PIPELINE=int(aff_min != 512)):
                    assert np_to_pt(*fb.run(s)) == exp, (c, aff_min, pairs_max)
                    assert np_to_pt(*fb.run(s[lo:hi], lo, hi)) == exp_part, (c, aff_min, pairs_max)
                    assert gpu_msm(dvp, s, bases) == exp, (c, aff_min, pairs_max)
                if pairs_max == 0:
                    # round 6: the reducer's first level without exceptional branches + its rest list (k_accum_affine_fast / _rest), its last
                    # level on rows of 16 lanes, ld_add_nodbl's doubling from a copy, the one-shot tail's group sums -- each switched off in turn
                    for knob in ("DVP_MSM_ACCUM_FAST", "DVP_MSM_ACCUM_HEX_MAX", "DVP_MSM_TAIL_GROUPS"):
                        with dvp.tune(DVP_MSM_AFF_MIN=aff_min, DVP_MSM_BUCKET_PAIRS_MAX=0, DVP_MSM_C=c, **{knob: 0}):
                            assert np_to_pt(*fb.run(s)) == exp, (c, aff_min, knob)
                            assert gpu_msm(dvp, s, bases) == exp, (c, aff_min, knob)
        fb.close()


def test_merge_tree_equal_and_opposite_buckets(dvp):
    """the merge tree's exceptional additions (k_merge, lambda-projective: equal operands are doubled from a copy read back,
    opposite ones give infinity): bases P, P, Q, -Q with scalars 2, 3, 4, 5 put the SAME point into buckets 2 and 3 and opposite
    points into buckets 4 and 5, which level 0 adds -- with few buckets (one quad per addition) and with 2^17 (one lane per
    addition); then the same with the pairs one level up (scalars 4, 6 / 8, 10: buckets (4,5)+(6,7) and (8,9)+(10,11))"""
    kp, kq = 0x1234567, 0x7654321
    P, Q = co.k233_mulgen(kp), co.k233_mulgen(kq)
    bases = pts_to_np([P, P, Q, o.k233_neg(Q)])
    for sv in ([2, 3, 4, 5], [4, 6, 8, 10], [1, 1, 3, 3], [6, 7, 1 << 100, (1 << 100) + 1]):
        exp = co.k233_mulgen(((sv[0] + sv[1]) * kp + (sv[2] - sv[3]) * kq) % o.P)
        for c in (8, 18):
            with dvp.tune(DVP_MSM_FIXED_C=c):
                fb = dvp.curve.FixedBaseMsm(bases)
            assert np_to_pt(*fb.run(to_limbs(sv))) == exp, (sv, c)
            fb.close()
        for c in (4, 12):
            with dvp.tune(DVP_MSM_C=c):
                assert gpu_msm(dvp, to_limbs(sv), bases) == exp, (sv, c)


def test_points_add_like_curvepoint_add(dvp):
    """CurvePoint::add (src/curve.rs:84-90): generic, doubling, P + (-P), neutral on either side -- vs the oracle group law"""
    import pyref as o2
    rnd = random.Random(44)
    ks = [rnd.randrange(1, o.P) for _ in range(6)]
    pts = [co.k233_mulgen(k) for k in ks]
    neg = lambda p: (p[0], p[0] ^ p[1])
    a = [pts[0], pts[1], pts[2], pts[3], pts[4], pts[5]]
    b = [pts[1], pts[1], neg(pts[2]), pts[0], pts[4], pts[3]]
    a_inf = np.array([0, 0, 0, 1, 0, 1], dtype=np.uint8)
    b_inf = np.array([0, 0, 0, 0, 1, 1], dtype=np.uint8)
    xy, inf = dvp.curve.add(pts_to_np(a), pts_to_np(b), a_inf, b_inf)
    exp = [o2.k233_add(a[0], b[0]), o2.k233_add(a[1], a[1]), None, b[3], a[4], None]
    for i, e in enumerate(exp):
        assert np_to_pt(xy[i], bool(inf[i])) == e, i


@pytest.mark.parametrize("log_n", [20, 22])
def test_msm_full_size_dlog_and_linearity(dvp, log_n):
    """full size, oracle-backed: bases k_i*G, so MSM(s) must equal (sum s_i k_i)*G computed by the OpenSSL-pinned C
    oracle -- for the one-shot and the fixed-base path (the shape of the reference's test_msm, src/curve.rs:218-232);
    plus the reference's linearity property MSM(s) + MSM(t) == MSM(s + t) (src/curve.rs:198-215)"""
    n = 1 << log_n
    k = rand_fr_np(n, 81)
    bases, _ = dvp.curve.point_scalar_mul_gen_batch(k)
    s, t = rand_fr_np(n, 82), rand_fr_np(n, 83)
    fb = dvp.curve.FixedBaseMsm(bases)
    exp_s = co.k233_mulgen(np_dot_mod_fast(s, k))
    ps_one, ps_fix = gpu_msm(dvp, s, bases), np_to_pt(*fb.run(s))
    assert ps_one == exp_s and ps_fix == exp_s
    if log_n == 20:
        exp_t = co.k233_mulgen(np_dot_mod_fast(t, k))
        assert np_to_pt(*fb.run(t)) == exp_t
        st_sum = to_limbs([(a + b) % o.P for a, b in zip(from_limbs(s), from_limbs(t))])
        pst = np_to_pt(*fb.run(st_sum))
        assert o.k233_add(exp_s, exp_t) == pst == gpu_msm(dvp, st_sum, bases)
    fb.close()


def test_fixed_base_vs_one_shot_randomised(dvp):
    """differential sweep: every fixed-base window size 8..21 of the signed and 8..20 of the tau-adic aligned windows (both sort
    flavours, 4/8/16-slot pair rounds) against the one-shot path on random sub-ranges, plus scalars with long runs of
    zero / one digits"""
    rnd = random.Random(2026)
    n = 3000
    k = rand_fr_np(n, 91)
    bases, _ = dvp.curve.point_scalar_mul_gen_batch(k)
    s = rand_fr_np(n, 92)
    special = [0, 1, 2, o.P - 1, o.P - 2, (1 << 231) - 1, 1 << 230, (1 << 116) + 1, 0xFFFFFFFFFFFFFFFF, (o.P - 1) // 2, 3 << 200]
    # worst cases of the tau-adic recoding: corners of the reduction's fundamental region (longest expansions, 236 of the
    # proven 240 digits) and the rounding boundaries of its fixed-point quotients (tau.cuh)
    special += tau_adversarial_scalars()
    assert max(len(co.tau_digits(x)) for x in special) >= 236
    s[: len(special)] = to_limbs(special)
    ks, ss = from_limbs(k), from_limbs(s)
    # slide: 0 = aligned windows of signed binary digits (the default flavour), 3 = aligned tau-adic windows
    for c, slide in [(c, 0) for c in range(8, 22)] + [(c, 3) for c in range(8, 21)]:
        with dvp.tune(DVP_MSM_FIXED_C=c, DVP_MSM_ALIGNED_SIGNED=0 if slide == 3 else 1,
                      DVP_MSM_AFF_MIN=rnd.choice([16, 256, 4096, 1 << 19]),
                      DVP_MSM_AFF_BMAX=rnd.choice([2, 7, 48]), DVP_MSM_SORT_FUSED=c & 1, DVP_MSM_ROUND_PIPELINE=(c >> 1) & 1):
            fb = dvp.curve.FixedBaseMsm(bases)
            if slide == 0:  # 234 bits in ceil(234 / c) windows evened out; all windows narrow = the plan of c - 1 (19 -> 18, 21 -> 20)
                ce = c
                while (234 + ce - 1) // ce * ce - 234 >= (234 + ce - 1) // ce:
                    ce -= 1
                assert fb.plan() == (ce, (234 + ce - 1) // ce)
            else:
                assert fb.plan() == (c, (234 + c - 1) // c + 1)
            for _ in range(3):
                lo = rnd.randrange(0, n - 1)
                hi = rnd.randrange(lo + 1, n + 1)
                xy, is_inf = fb.run(s[lo:hi], lo, hi)
                exp = co.k233_mulgen(sum(ss[i] * ks[i] for i in range(lo, hi)) % o.P)
                assert np_to_pt(xy, is_inf) == exp, (c, slide, lo, hi)
                assert gpu_msm(dvp, s[lo:hi], bases[lo:hi]) == exp, (c, slide, lo, hi)
            fb.close()


def test_one_shot_randomised_shapes_and_knobs(dvp):
    """differential sweep of the ONE-SHOT path (per-window bucket sets): random sizes 1..40 000, scalar populations
    (uniform / tiny / few distinct values / zeros and p-1 mixed in), neutral bases, every window size the cost model can
    pick (4..15), several reducer fan-ins, pair rounds all the way down and switched off -- each against the
    discrete-log identity with the OpenSSL-pinned oracle"""
    rnd = random.Random(4242)
    nmax = 40000
    k = rand_fr_np(nmax, 191)
    bases, _ = dvp.curve.point_scalar_mul_gen_batch(k)
    ks = from_limbs(k)
    for trial in range(28):
        n = rnd.choice([1, 2, 3, 63, 64, 65, 1000, 4097, rnd.randrange(1, nmax)])
        lo = rnd.randrange(0, nmax - n + 1)
        kind = trial % 4
        if kind == 0:
            sv = [rnd.randrange(o.P) for _ in range(n)]
        elif kind == 1:
            sv = [rnd.randrange(0, 16) for _ in range(n)]
        elif kind == 2:
            vals = [rnd.randrange(o.P) for _ in range(3)]
            sv = [rnd.choice(vals) for _ in range(n)]
        else:
            sv = [rnd.choice([0, o.P - 1, 1, rnd.randrange(o.P)]) for _ in range(n)]
        inf = np.zeros(n, dtype=np.uint8)
        if n > 4 and trial % 3 == 0:
            inf[rnd.randrange(n)] = 1
        exp = co.k233_mulgen(sum(s * ks[lo + i] for i, s in enumerate(sv) if not inf[i]) % o.P)
        knobs = dict(DVP_MSM_C=rnd.choice([0, 4, 7, 11, 15]), DVP_MSM_K=rnd.choice([0, 2, 5, 64]),
                     DVP_MSM_AFF_MIN=rnd.choice([1, 64, 1 << 19]), DVP_MSM_PROJ=int(trial % 7 == 6), DVP_MSM_AFF_BMAX=rnd.choice([1, 5, 48]))
        with dvp.tune(**knobs):
            assert gpu_msm(dvp, to_limbs(sv), bases[lo:lo + n], inf) == exp, (trial, n, knobs)


@pytest.mark.parametrize("c", [8, 13, 18, 19, 20, 21])
def test_signed_recode_words_vs_restatement(dvp, c):
    """k_recode_signed (the small-table flavour: aligned windows of signed binary digits) word for word against the textbook
    signed-digit decomposition restated here: 234 bits in W = ceil(234 / c) windows, the W c - 234 LOW windows one bit
    narrower (so no window is short: a short top window would pile every scalar into a handful of buckets); window w holds
    its bits plus the carry from below, a digit above half its range becomes digit - 2^width with a carry;
    sum_w d_w 2^(o_w) is the scalar and |d_w| <= 2^(width_w - 1).  A c whose windows would all be narrow is refused
    (it is the plan of c - 1)."""
    import ctypes as C

    rnd = random.Random(500 + c)
    W = C.c_int(0)
    n_win = (234 + c - 1) // c
    n_narrow = n_win * c - 234
    rc = dvp.lib.dvp_debug_recode_signed(None, 0, c, None, C.byref(W))
    if n_narrow >= n_win:
        assert rc == -1
        return
    dvp.check(rc, "windows")
    assert W.value == n_win
    widths = [c - 1 if w < n_narrow else c for w in range(n_win)]
    assert sum(widths) == 234
    vals = [0, 1, 2, (1 << c) - 1, 1 << (c - 1), (1 << (c - 1)) + 1, (1 << (c - 2)), (1 << (c - 2)) + 1, o.P - 1, (1 << 231) - 1, 1 << 230,
            int("1" * 231, 2)] + [rnd.randrange(o.P) for _ in range(1500)] + [rnd.randrange(1 << rnd.randrange(1, 232)) for _ in range(300)]
    # every window at exactly half its range, and one past it (the wrap / carry boundary of each width)
    off = 0
    for wd in widths[:-1]:
        vals += [1 << (off + wd - 1), (1 << (off + wd - 1)) + (1 << off), ((1 << wd) - 1) << off]
        off += wd
    s = to_limbs(vals)
    words = np.zeros((W.value, len(vals)), dtype=np.uint32)
    dvp.check(dvp.lib.dvp_debug_recode_signed(s.ctypes.data, len(vals), c, words.ctypes.data, C.byref(W)), "recode")
    for i, x in enumerate(vals):
        exp, carry, off = [], 0, 0
        for wd in widths:
            d = ((x >> off) & ((1 << wd) - 1)) + carry
            carry = 0
            if d > (1 << (wd - 1)):
                d, carry = d - (1 << wd), 1
            exp.append((off, d))
            off += wd
        assert carry == 0 and sum(d << o_w for o_w, d in exp) == x
        for w, (_, d) in enumerate(exp):
            word = int(words[w, i])
            if d == 0:
                assert word == 0, (c, i, w)
            else:
                assert word >> 31 == 1 and ((word >> 20) & 0xFF) == w and bool(word & 0x10000000) == (d < 0), (c, i, w, hex(word))
                assert (word & 0xFFFFF) == (abs(d) & ((1 << (c - 1)) - 1)), (c, i, w, d, hex(word))  # bucket key; 2^(c-1) -> bucket 0
