#!/usr/bin/env python3
"""Pin the 30-byte xsk233 point encoding and the generator convention against ONE vector from xs233.

The reference reaches its curve through Thomas Pornin's xs233 (xs233-sys =0.2.0, src/curve.rs:13,93-109); that
source is not available offline and the reference holds no known-answer bytes, so the codec rule of this repo
(dv-pari_amd/csrc/codec.hip: k_encode / k_decode) is a CANDIDATE.  A maintainer with cargo can close the question
with one line of Rust in the reference tree,

    let p = point_scalar_mul_gen(Fr::from(k));  println!("{}", hex::encode(p.to_bytes()));      // src/curve.rs:93-100,129-137

and then

    python tools/pin_xsk233.py <k as hex> <the 60 hex digits printed>

This script is self-contained (its own GF(2^233) / K-233 arithmetic in python ints; it does not import the oracle).
It recomputes k*G for each generator convention, applies every candidate encoding rule, and names the combination
that reproduces the bytes -- "CURRENT RULE CONFIRMED" if it is the one codec.hip implements, otherwise the name of the
rule NUMBER to hand to dvp_codec_set_rule (include/dvpari.h; every candidate of the family is already implemented in codec.hip
and in oracle/pyref.py:xsk233_encode(pt, rule), parity-tested rule by rule).  With --gpu it also checks that the library's own
dvp_mulgen_batch (k_mulgen + k_encode) emits the same 30 bytes under that rule.

    python tools/pin_xsk233.py --self-test       # encodes with the current rule and finds it again
"""
import sys

M = 233
POLY = (1 << 233) | (1 << 74) | 1
ORDER = 0x8000000000000000000000000000069D5BB915BCD46EFB1AD5F173ABDF
GX = 0x017232BA853A7E731AF129F22FF4149563A419C26BF50A4C9D6EEFAD6126
GY = 0x01DB537DECE819B7F70F555A67C427A8CD9BF18AEB9B56E0C11056FAE6A3


def gf_red(c):
    for i in range(c.bit_length() - 1, M - 1, -1):
        if (c >> i) & 1:
            c ^= POLY << (i - M)
    return c


def gf_mul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    return gf_red(r)


def gf_sqr(a):
    return gf_mul(a, a)


def gf_inv(a):
    assert a
    r, e = 1, (1 << M) - 2
    while e:
        if e & 1:
            r = gf_mul(r, a)
        a = gf_sqr(a)
        e >>= 1
    return r


def gf_sqrt(a):
    for _ in range(M - 1):
        a = gf_sqr(a)
    return a


# y^2 + xy = x^3 + 1 (a = 0, b = 1); None = infinity
def add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if y1 != y2 or x1 == 0:
            return None
        lam = x1 ^ gf_mul(y1, gf_inv(x1))
        x3 = gf_sqr(lam) ^ lam
    else:
        lam = gf_mul(y1 ^ y2, gf_inv(x1 ^ x2))
        x3 = gf_sqr(lam) ^ lam ^ x1 ^ x2
    return (x3, gf_mul(lam, x1 ^ x3) ^ x3 ^ y1)


def mul(k, p):
    r = None
    while k:
        if k & 1:
            r = add(r, p)
        p = add(p, p)
        k >>= 1
    return r


N = (0, 1)  # the point of order 2; xsk233 = { P + N : P in E[r] }

# ---- candidate rules: value in GF(2^233) computed from a curve point (x, y) of the standard model ---------------------
def _shifted(pt):  # model with N at (0,0): y' = y + sqrt(b) = y + 1
    return pt[0], pt[1] ^ 1


def rule_sqrt_s_over_x(pt):  # CURRENT (codec.hip): s = y' + x^2 + a x + sqrt(b), w = sqrt(s / x)
    x, yp = _shifted(pt)
    return gf_sqrt(gf_mul(yp ^ gf_sqr(x) ^ 1, gf_inv(x)))


def rule_s_over_x(pt):
    x, yp = _shifted(pt)
    return gf_mul(yp ^ gf_sqr(x) ^ 1, gf_inv(x))


def rule_yp_over_x(pt):
    x, yp = _shifted(pt)
    return gf_mul(yp, gf_inv(x))


def rule_sqrt_yp_over_x(pt):
    return gf_sqrt(rule_yp_over_x(pt))


def rule_lambda(pt):  # x + y/x on the standard model
    x, y = pt
    return x ^ gf_mul(y, gf_inv(x))


def rule_sqrt_lambda(pt):
    return gf_sqrt(rule_lambda(pt))


def rule_y_over_x(pt):
    x, y = pt
    return gf_mul(y, gf_inv(x))


def rule_x(pt):
    return pt[0]


RULES = [("sqrt(s/x), s = y'+x^2+ax+sqrt(b)  [current codec.hip rule]", rule_sqrt_s_over_x), ("s/x", rule_s_over_x),
         ("y'/x", rule_yp_over_x), ("sqrt(y'/x)", rule_sqrt_yp_over_x), ("lambda = x + y/x", rule_lambda),
         ("sqrt(lambda)", rule_sqrt_lambda), ("y/x", rule_y_over_x), ("x alone", rule_x)]
VIEWS = [("of P + N (the xsk233 element)", lambda p: add(p, N)), ("of P (the E[r] representative)", lambda p: p)]
GENS = [("generator = G_std + N", (GX, GY)), ("generator = -G_std + N", (GX, GX ^ GY))]
CURRENT = (0, 0, 0, 0, "little")


def candidates(k):
    for gi, (gname, g) in enumerate(GENS):
        p = mul(k % ORDER, g)
        if p is None:
            continue
        for vi, (vname, view) in enumerate(VIEWS):
            q = view(p)
            if q is None or q[0] == 0:
                continue
            for ri, (rname, rule) in enumerate(RULES):
                w = rule(q)
                for plus1 in (0, 1):
                    for order in ("little", "big"):
                        yield (gi, vi, ri, plus1, order), (w ^ plus1).to_bytes(30, order), f"{rname} {vname}{' + 1' if plus1 else ''}; {gname}; {order}-endian"


def encode_current(k):
    for key, enc, _ in candidates(k):
        if key == CURRENT:
            return enc
    raise AssertionError


TRANSFORM = {0: 0, 1: 1, 2: 0, 3: 2, 4: 1, 5: 0}  # RULES index -> codec.hip transform (0: w, 1: w^2, 2: sqrt w); y/x and x: none


def rule_number(key):
    """the dvp_codec_set_rule number of a candidate (None for the two formulas outside the family): the view, the generator's
    sign and the explicit "+1" are one parity bit (w(Q + N) = w(-Q) = w(Q) + 1, and squaring / square root commute with + 1)"""
    gi, vi, ri, plus1, order = key
    if ri not in TRANSFORM:
        return None
    return (gi ^ vi ^ plus1) | (2 if order == "big" else 0) | (TRANSFORM[ri] << 2)


def pin(k, target: bytes, out=print):
    """several candidates are the same function written differently (on the curve sqrt(s/x) = y'/x = sqrt(lambda);
    w(Q + N) = w(Q) + 1 = w(-Q)), so the matches are reported as ONE class of equivalent forms"""
    hits = [(key, name) for key, enc, name in candidates(k) if enc == target]
    if not hits:
        out("NO CANDIDATE MATCHES -- the rule is outside this script's list; the bytes to reproduce are " + target.hex())
        return 2
    current = any(key == CURRENT for key, _ in hits)
    out("CURRENT RULE CONFIRMED (codec.hip k_encode / k_decode and the generator convention are right):" if current
        else "MATCH, BUT NOT THE RULE codec.hip IMPLEMENTS -- port this rule (any of its equivalent forms):")
    for key, name in hits:
        out(("  * " if key == CURRENT else "    ") + name)
    nums = sorted({rule_number(key) for key, _ in hits if rule_number(key) is not None})
    if len(nums) == 1:
        out(f"=> dvp_codec_set_rule({nums[0]})   (or DVP_CODEC_RULE={nums[0]} in the environment; oracle: pyref.xsk233_encode(pt, {nums[0]}))"
            + ("   -- the default" if nums[0] == 0 else ""))
    elif nums:
        out(f"=> ambiguous between rules {nums}: supply a second vector")
    else:
        out("=> outside the family codec.hip can be switched to (y/x or x alone): a port is needed")
    return 0 if current else 1


def main(argv):
    if "--self-test" in argv:
        k = 0x1234567890ABCDEF1234567890ABCDEF
        enc = encode_current(k)
        print("current rule encodes", hex(k), "* G as", enc.hex())
        return pin(k, enc)
    args = [a for a in argv if not a.startswith("--")]
    if len(args) != 2:
        print(__doc__)
        return 64
    k, target = int(args[0], 16), bytes.fromhex(args[1])
    if len(target) != 30:
        print("expected 30 bytes (60 hex digits)")
        return 64
    rc = pin(k, target)
    if "--gpu" in argv:
        import importlib
        import os

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        dvp = importlib.import_module("dv-pari_amd")
        nums = sorted({rule_number(key) for key, enc, _ in candidates(k) if enc == target and rule_number(key) is not None})
        if len(nums) == 1:
            dvp.check(dvp.lib.dvp_codec_set_rule(nums[0]))
        got = dvp.curve.point_scalar_mul_gen_batch_bytes(dvp.fr.vec([k]))[0].tobytes()
        print("library (k_mulgen + k_encode):", got.hex(), "== supplied vector" if got == target else "!= supplied vector")
        rc = rc or (0 if got == target else 1)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
