"""
ORACLE (test infrastructure, NOT product code) -- pure-Python big-int restatement of the
DV-Pari prover hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file; the product path (dv-pari_amd/) never does.

Every function cites the reference file:line it restates (paths relative to the upstream
alpenlabs/dv-pari tree).  The arithmetic of the hot path lives in un-vendored third-party
crates (xs233-sys =0.2.0, alpenlabs/ecfft@9c6cac7, ark-ff 0.5.0, blake3 1.8.2); their
published algorithms are restated here from the mathematics:

  * Fr            : integers mod p (src/curve.rs:16-22)
  * GF(2^233)     : GF(2)[z]/(z^233+z^74+1), Python ints as bit vectors
  * K-233         : y^2+xy = x^3+1 over GF(2^233), affine group law; pinned against OpenSSL
                    sect233k1 vectors in tests/golden/k233_openssl.json
  * xsk233 codec  : 30-byte encoding of the prime-order group E[r]+N (Pornin, ePrint 2022/1325).
                    PARITY UNPINNED: the reference holds no known-answer bytes and the xs233
                    source is not available offline; the candidate rule is isolated in
                    xsk233_encode/xsk233_decode below.
  * ECFFT         : Ben-Sasson/Carmon/Kopparty/Levit part I; domain constants from
                    src/ec_fft.rs:205-229; extend/enter/exit outputs are mathematically unique
                    and are pinned against O(n^2) Lagrange interpolation (reference test
                    src/ec_fft.rs:883-907).
  * BLAKE3        : from the BLAKE3 spec; pinned by the official empty-input vector.
"""
from __future__ import annotations

import struct

# --------------------------------------------------------------------------------------------
# Fr  (src/curve.rs:16-22)
# --------------------------------------------------------------------------------------------
P = 3450873173395281893717377931138512760570940988862252126328087024741343
FR_BITS = 232


def fr_inv(a: int) -> int:
    return pow(a % P, P - 2, P)


def fr_batch_inverse(v):
    """ark_ff::batch_inversion semantics (zeros are left as zeros)."""
    out = [0] * len(v)
    acc = 1
    pref = []
    for x in v:
        pref.append(acc)
        if x % P:
            acc = acc * x % P
    inv = fr_inv(acc)
    for i in range(len(v) - 1, -1, -1):
        if v[i] % P:
            out[i] = inv * pref[i] % P
            inv = inv * v[i] % P
    return out


def fr_to_le_bytes_stripped(x: int) -> bytes:
    """src/curve.rs:162-182 -- canonical LE, truncated to 30 B, trailing zero bytes stripped."""
    b = (x % P).to_bytes(32, "little")[:30]
    return b.rstrip(b"\x00")


def frbits_from_fr(x: int):
    """src/curve.rs:30-40 -- 232 LE bits."""
    return [(x >> i) & 1 for i in range(232)]


def frbits_to_fr(bits):
    """src/curve.rs:43-59 -- returns (value, is_valid)."""
    n = 0
    for i, b in enumerate(bits):
        if b:
            n |= 1 << i
    if n >= P:
        return 0, False
    return n, True


# --------------------------------------------------------------------------------------------
# GF(2^233) = GF(2)[z]/(z^233 + z^74 + 1)
# --------------------------------------------------------------------------------------------
GF_M = 233
GF_POLY = (1 << 233) | (1 << 74) | 1
GF_MASK = (1 << 233) - 1


def gf_reduce(c: int) -> int:
    while c >> GF_M:
        hi = c >> GF_M
        c = (c & GF_MASK) ^ hi ^ (hi << 74)
    return c


def gf_clmul(a: int, b: int) -> int:
    r = 0
    while b:
        lsb = b & -b
        r ^= a * lsb  # a << tz(b)
        b ^= lsb
    return r


def gf_mul(a: int, b: int) -> int:
    return gf_reduce(gf_clmul(a, b))


def gf_sqr(a: int) -> int:
    return gf_mul(a, a)


def gf_pow2k(a: int, k: int) -> int:
    for _ in range(k):
        a = gf_sqr(a)
    return a


def gf_inv(a: int) -> int:
    """Itoh-Tsujii: a^(2^233-2).  Chain on 232 = 11101000b."""
    assert a != 0
    # b_k = a^(2^k - 1)
    b1 = a
    b2 = gf_mul(gf_pow2k(b1, 1), b1)
    b3 = gf_mul(gf_pow2k(b2, 1), b1)
    b6 = gf_mul(gf_pow2k(b3, 3), b3)
    b7 = gf_mul(gf_pow2k(b6, 1), b1)
    b14 = gf_mul(gf_pow2k(b7, 7), b7)
    b28 = gf_mul(gf_pow2k(b14, 14), b14)
    b29 = gf_mul(gf_pow2k(b28, 1), b1)
    b58 = gf_mul(gf_pow2k(b29, 29), b29)
    b116 = gf_mul(gf_pow2k(b58, 58), b58)
    b232 = gf_mul(gf_pow2k(b116, 116), b116)
    return gf_sqr(b232)


def gf_sqrt(a: int) -> int:
    return gf_pow2k(a, 232)


def gf_trace(a: int) -> int:
    # Tr(z^i) = 1 only for i in {0, 159} for z^233+z^74+1 (checked in tests against the definition)
    return ((a >> 0) ^ (a >> 159)) & 1


def gf_trace_def(a: int) -> int:
    t, x = 0, a
    for _ in range(GF_M):
        t ^= x
        x = gf_sqr(x)
    assert t in (0, 1)
    return t


def gf_halftrace(c: int) -> int:
    """H(c) = sum_{i=0}^{116} c^(2^(2i)); solves z^2+z=c when Tr(c)=0 (m odd)."""
    h, x = 0, c
    for _ in range(117):
        h ^= x
        x = gf_sqr(gf_sqr(x))
    return h


def gf_solve_quadratic(c: int):
    """Return z with z^2+z=c, or None when Tr(c)=1."""
    if gf_trace(c):
        return None
    z = gf_halftrace(c)
    assert gf_sqr(z) ^ z == c
    return z


# --------------------------------------------------------------------------------------------
# K-233 / sect233k1:  y^2 + xy = x^3 + a x^2 + B,  a = 0, B = 1.  Affine, None = infinity.
# --------------------------------------------------------------------------------------------
K233_A = 0
K233_B = 1
K233_ORDER = P  # prime subgroup order r == Fr modulus (src/ec_fft.rs:41)
K233_COFACTOR = 4
G_STD = (
    0x017232BA853A7E731AF129F22FF4149563A419C26BF50A4C9D6EEFAD6126,
    0x01DB537DECE819B7F70F555A67C427A8CD9BF18AEB9B56E0C11056FAE6A3,
)
N_STD = (0, 1)  # the point of order 2: (0, sqrt(B))


def k233_on_curve(pt) -> bool:
    if pt is None:
        return True
    x, y = pt
    return gf_sqr(y) ^ gf_mul(x, y) == gf_mul(gf_sqr(x), x) ^ gf_mul(K233_A, gf_sqr(x)) ^ K233_B


def k233_neg(pt):
    if pt is None:
        return None
    return (pt[0], pt[0] ^ pt[1])


def k233_dbl(pt):
    if pt is None:
        return None
    x, y = pt
    if x == 0:
        return None
    lam = x ^ gf_mul(y, gf_inv(x))
    x3 = gf_sqr(lam) ^ lam ^ K233_A
    y3 = gf_sqr(x) ^ gf_mul(lam ^ 1, x3)
    return (x3, y3)


def k233_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if y1 == y2:
            return k233_dbl(p1)
        return None
    lam = gf_mul(y1 ^ y2, gf_inv(x1 ^ x2))
    x3 = gf_sqr(lam) ^ lam ^ x1 ^ x2 ^ K233_A
    y3 = gf_mul(lam, x1 ^ x3) ^ x3 ^ y1
    return (x3, y3)


def k233_mul(k: int, pt):
    k %= K233_ORDER * K233_COFACTOR
    r = None
    q = pt
    while k:
        if k & 1:
            r = k233_add(r, q)
        q = k233_dbl(q)
        k >>= 1
    return r


def k233_msm(scalars, points):
    """src/curve.rs:141-158 -- sum of independent scalar multiplications (as a group element)."""
    assert len(scalars) == len(points)
    acc = None
    for s, pt in zip(scalars, points):
        acc = k233_add(acc, k233_mul(s % P, pt))
    return acc


def k233_in_prime_subgroup(pt) -> bool:
    return k233_mul(K233_ORDER, pt) is None


def k233_in_prime_subgroup_fast(pt) -> bool:
    """pt in E[r] <=> pt in 4E: Tr(x)=Tr(a) and, for a half Q of pt, Tr(x_Q)=Tr(a)."""
    if pt is None:
        return True
    x, y = pt
    if x == 0:
        return False
    if gf_trace(x) != gf_trace(K233_A):
        return False
    lam = gf_solve_quadratic(x ^ K233_A)
    u2 = y ^ gf_mul(lam ^ 1, x)  # x_Q^2 for one of the two halves
    return gf_trace(u2) == gf_trace(K233_A)


# --------------------------------------------------------------------------------------------
# xsk233 group codec -- CANDIDATE RULE, PARITY UNPINNED (see header).
#
# Group = { P + N : P in E[r] } with law (P+N) (+) (Q+N) = P+Q+N, neutral N.  Internally every
# element is carried as its E[r] representative P ("internal" form), so that
# xsk233 MSM == plain K-233 MSM on the representatives (src/curve.rs:141-158 is linear).
# Model with N=(0,0): y' = y + sqrt(B);  s = y' + x^2 + a x + b (b = sqrt(B));  w = sqrt(s/x);
# encode(N) = 0.  30 bytes little-endian, top 7 bits zero.
# --------------------------------------------------------------------------------------------
XSK_B = 1  # sqrt(B)


XSK_RULES = 12  # rule = bit 0: "+1" (the E[r] representative's own value) | bit 1: big-endian bytes | bits 2..3: 0 w, 1 w^2, 2 sqrt(w)


def xsk233_encode(p_internal, rule: int = 0) -> bytes:
    """Restates CurvePoint::to_bytes -> xsk233_encode (src/curve.rs:93-100).  `rule` selects which equivalent presentation of
    the encoded field element the bytes hold (codec.hip: dvp_codec_set_rule; 0 = the candidate followed by default); every
    candidate tools/pin_xsk233.py enumerates is one of them, each computed HERE from its own defining formula."""
    if p_internal is None:
        return bytes(30)
    assert 0 <= rule < XSK_RULES
    q = k233_add(p_internal, N_STD)
    assert q is not None, "N itself is not an element of E[r]"
    view = p_internal if rule & 1 else q  # w(Q + N) = w(Q) + 1: the "+1" rules are the value of the E[r] representative itself
    x, y = view
    yp = y ^ 1
    s = yp ^ gf_sqr(x) ^ gf_mul(K233_A, x) ^ XSK_B
    s_over_x = gf_mul(s, gf_inv(x))
    tr = (rule >> 2) & 3
    if tr == 0:
        t = gf_sqrt(s_over_x)                     # w = sqrt(s/x)  (= y'/x = sqrt(lambda))
    elif tr == 1:
        t = s_over_x                              # s/x (= lambda = x + y/x on the standard model)
        assert t == x ^ gf_mul(y, gf_inv(x))
    else:
        t = gf_sqrt(gf_mul(yp, gf_inv(x)))        # sqrt(y'/x)
    return t.to_bytes(30, "big" if rule & 2 else "little")


def xsk233_decode(buf: bytes, rule: int = 0):
    """Restates CurvePoint::from_bytes -> xsk233_decode (src/curve.rs:103-109).
    Returns (internal point, ok)."""
    assert len(buf) == 30 and 0 <= rule < XSK_RULES
    w = int.from_bytes(buf, "big" if rule & 2 else "little")
    if w >> 233:
        return None, False
    if w == 0:
        return None, True  # neutral N  -> internal infinity
    if rule & 1:
        w ^= 1
    tr = (rule >> 2) & 3
    if tr == 1:
        w = gf_sqrt(w)
    elif tr == 2:
        w = gf_sqr(w)
    e = gf_sqr(w) ^ w ^ K233_A
    if e == 0:
        return None, False
    # x^2 + e x + b = 0  ->  x = e z,  z^2 + z = b / e^2
    z = gf_solve_quadratic(gf_mul(XSK_B, gf_inv(gf_sqr(e))))
    if z is None:
        return None, False
    for zz in (z, z ^ 1):
        x = gf_mul(e, zz)
        if x == 0:
            continue
        s = gf_mul(gf_sqr(w), x)
        yp = s ^ gf_sqr(x) ^ gf_mul(K233_A, x) ^ XSK_B
        q = (x, yp ^ 1)
        if not k233_on_curve(q):
            continue
        p_int = k233_add(q, N_STD)
        if k233_in_prime_subgroup_fast(p_int):
            return p_int, True
    return None, False


XSK233_GENERATOR_INTERNAL = G_STD  # xsk233_generator ~ G_std + N (UNVERIFIED, see header)


# --------------------------------------------------------------------------------------------
# ECFFT over Fr  (domain constants: src/ec_fft.rs:205-229)
# --------------------------------------------------------------------------------------------
ECFFT_A = 2125753088427212854352924174339172498722499297750753614229533284661082
ECFFT_B = 3303427382072851929105738691313541325219445842218525662544269869787589
ECFFT_GEN = (
    1969398527398874941115360315313056361667745675958024267654083765592400,
    917696706299601920847965073366118878832337776859300472447868491055982,
)
ECFFT_COSET = (
    1557215852494830750811239888869886110709986867282698163663807961412586,
    2302954593454110051167704558708330032236229062988890422530712548754008,
)
ECFFT_LOG_ORDER = 28  # subgroup_adic, src/ec_fft.rs:205


def sw_add(p1, p2, a):
    """Short-Weierstrass affine addition over Fr (ecfft::ec::Point +)."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1 + a) * fr_inv(2 * y1) % P
    else:
        lam = (y2 - y1) * fr_inv(x2 - x1) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def sw_mul(k, pt, a):
    r, q = None, pt
    while k:
        if k & 1:
            r = sw_add(r, q, a)
        q = sw_add(q, q, a)
        k >>= 1
    return r


class FFTree:
    """
    Leaves + isogeny chain for a 2^log_n-leaf ECFFT domain (build_ec_fftrees,
    src/ec_fft.rs:93-170).  layers[d][i] = x-coordinate leaves at depth d; x0[d], t[d] define
    psi_d(x) = x + t/(x - x0) (Velu, kernel <(x0,0)>) which is the 2-isogeny lowering the
    2-adicity of the generator (src/ec_fft.rs:131-148).
    `stride` views: the sub-tree of src/ec_fft.rs:21-25 (even leaves at every layer).
    """

    def __init__(self, log_n: int, shifted: bool = False, base_log_n: int | None = None):
        assert 1 <= log_n <= ECFFT_LOG_ORDER
        if base_log_n is None:
            base_log_n = log_n
        self.log_n = log_n
        n = 1 << log_n
        a = ECFFT_A
        g = ECFFT_GEN
        for _ in range(ECFFT_LOG_ORDER - log_n):
            g = sw_add(g, g, a)
        coset = ECFFT_COSET
        if shifted:  # src/ec_fft.rs:151-155
            bg = ECFFT_GEN
            for _ in range(ECFFT_LOG_ORDER - base_log_n):
                bg = sw_add(bg, bg, a)
            coset = sw_add(coset, bg, a)
        # leaves (src/ec_fft.rs:158-162)
        leaves = []
        pt = coset
        for _ in range(n):
            leaves.append(pt[0])
            pt = sw_add(pt, g, a)
        # x-coordinates of the points of order 2^j in <g>, j = 1..log_n
        q = []
        h = g
        tmp = [h]
        for _ in range(log_n - 1):
            h = sw_add(h, h, a)
            tmp.append(h)
        # tmp[i] has order 2^(log_n - i); order 2^j point is tmp[log_n - j]
        for j in range(1, log_n + 1):
            q.append(tmp[log_n - j][0])
        assert tmp[log_n - 1][1] == 0  # order-2 point has y = 0
        self.layers = [leaves]
        self.x0 = []
        self.t = []
        self.a_chain = [a]
        cur = leaves
        for d in range(log_n):
            x0 = q[d]
            t = (3 * x0 * x0 + a) % P
            self.x0.append(x0)
            self.t.append(t)
            half = len(cur) // 2
            nxt = [(x + t * fr_inv(x - x0)) % P for x in cur[:half]]
            # consistency: psi(L[i]) == psi(L[i+half])
            if half:
                i = half - 1
                assert nxt[i] == (cur[i + half] + t * fr_inv(cur[i + half] - x0)) % P
            q = [None] * (d + 1) + [(x + t * fr_inv(x - x0)) % P for x in q[d + 1:]]
            a = (a - 5 * t) % P
            self.a_chain.append(a)
            self.layers.append(nxt)
            cur = nxt

    # -- views -------------------------------------------------------------------------------
    def leaves(self, stride: int = 1):
        return self.layers[0][::stride]

    def both_domains(self):
        """get_both_domains, src/ec_fft.rs:179-189."""
        l = self.layers[0]
        return l[0::2], l[1::2]

    # -- extend ------------------------------------------------------------------------------
    def extend(self, ev, stride: int = 1, to_even: bool = False):
        """
        FFTree::extend(evals, Moiety::S1) (call site src/proving.rs:412): n evaluations on the
        even leaves of the (strided) tree -> n evaluations on the odd leaves, for the unique
        interpolant of degree < n.  to_even=True is the mirrored direction (Moiety::S0), needed
        by exit.
        """
        return self._extend(list(ev), 0, stride, to_even)

    def _extend(self, ev, d, stride, to_even):
        n = len(ev)
        if n == 1:
            return ev[:]
        L = self.layers[d][::stride]
        assert len(L) == 2 * n
        h = n // 2
        e = h - 1
        x0 = self.x0[d]
        src, dst = (1, 0) if to_even else (0, 1)
        p0, p1 = [0] * h, [0] * h
        for i in range(h):
            s0, s1 = L[2 * i + src], L[2 * i + src + n]
            v0, v1 = pow(s0 - x0, e, P), pow(s1 - x0, e, P)
            # [e0;e1] = [[v0, s0 v0],[v1, s1 v1]] [p0;p1]
            t0 = ev[i] * fr_inv(v0) % P
            t1 = ev[i + h] * fr_inv(v1) % P
            dinv = fr_inv(s1 - s0)
            p1[i] = (t1 - t0) * dinv % P
            p0[i] = (s1 * t0 - s0 * t1) * dinv % P
        f0 = self._extend(p0, d + 1, stride, to_even)
        f1 = self._extend(p1, d + 1, stride, to_even)
        out = [0] * n
        for i in range(h):
            s0, s1 = L[2 * i + dst], L[2 * i + dst + n]
            v0, v1 = pow(s0 - x0, e, P), pow(s1 - x0, e, P)
            out[i] = v0 * (f0[i] + s0 * f1[i]) % P
            out[i + h] = v1 * (f0[i] + s1 * f1[i]) % P
        return out

    # -- enter / exit ------------------------------------------------------------------------
    def enter(self, coeffs, stride: int = 1):
        """FFTree::enter (call sites src/ec_fft.rs:317,411): coefficients -> evaluations on the
        leaves of the (strided) tree."""
        n = len(coeffs)
        L = self.layers[0][::stride]
        assert len(L) == n
        if n == 1:
            return [coeffs[0] % P]
        h = n // 2
        u0 = self.enter(coeffs[:h], stride * 2)
        v0 = self.enter(coeffs[h:], stride * 2)
        u1 = self.extend(u0, stride)
        v1 = self.extend(v0, stride)
        out = [0] * n
        for i in range(h):
            out[2 * i] = (u0[i] + pow(L[2 * i], h, P) * v0[i]) % P
            out[2 * i + 1] = (u1[i] + pow(L[2 * i + 1], h, P) * v1[i]) % P
        return out

    def vanish_even_at(self, x, stride: int = 1):
        """Z_0(x): vanishing polynomial of the even leaves of the strided tree, via the
        isogeny chain: the even leaves all map to layers[k][0] under psi_{k-1}..psi_0, and
        Z_0 = U - c V with (U,V) the projective image (U monic of degree 2^k)."""
        L = self.layers[0][::stride]
        m = len(L) // 2  # number of even leaves
        k = m.bit_length() - 1
        u, v = x % P, 1
        for d in range(k):
            x0, t = self.x0[d], self.t[d]
            u, v = (u * u - x0 * u * v + t * v * v) % P, (u - x0 * v) * v % P
        c = self.layers[k][0]
        return (u - c * v) % P

    def _redc_z0(self, ev, a_vals, stride):
        """<P * Z_0^{-1} mod A> on S from <P> on S; A given by its evaluations a_vals on S."""
        n = len(ev)
        L = self.layers[0][::stride]
        h = n // 2
        t0 = [ev[2 * i] * fr_inv(a_vals[2 * i]) % P for i in range(h)]
        g1 = self.extend(t0, stride)
        h1 = [
            (ev[2 * i + 1] - g1[i] * a_vals[2 * i + 1]) * fr_inv(self.vanish_even_at(L[2 * i + 1], stride)) % P
            for i in range(h)
        ]
        h0 = self.extend(h1, stride, to_even=True)
        out = [0] * n
        out[0::2] = h0
        out[1::2] = h1
        return out

    def exit(self, ev, stride: int = 1):
        """FFTree::exit (call site src/ec_fft.rs:266): evaluations on the leaves -> coefficients."""
        n = len(ev)
        if n == 1:
            return [ev[0] % P]
        L = self.layers[0][::stride]
        assert len(L) == n
        h = n // 2
        xnn = [pow(x, h, P) for x in L]
        c = self._z0z0_rem_xnn(stride)
        r1 = self._redc_z0(ev, xnn, stride)
        r1 = [r1[i] * c[i] % P for i in range(n)]
        u = self._redc_z0(r1, xnn, stride)  # <P mod X^h> on S
        u0 = u[0::2]
        lo = self.exit(u0, stride * 2)
        v0 = [(ev[2 * i] - u0[i]) * fr_inv(xnn[2 * i]) % P for i in range(h)]
        hi = self.exit(v0, stride * 2)
        return lo + hi

    def _z0z0_rem_xnn(self, stride):
        """<Z_0^2 mod X^h> on S (the table named z0z0_rem_xnn_s, src/tree_io.rs:34-48), computed
        here by brute force from the roots (oracle only)."""
        key = ("zz", stride)
        cache = self.__dict__.setdefault("_cache", {})
        if key in cache:
            return cache[key]
        L = self.layers[0][::stride]
        n = len(L)
        h = n // 2
        z = poly_from_roots(L[0::2])
        zz = poly_mul(z, z)[:h]
        out = [poly_eval(zz, x) for x in L]
        cache[key] = out
        return out


# -- small polynomial helpers (oracle only; O(n^2)) ---------------------------------------------
def poly_eval(c, x):
    r = 0
    for a in reversed(c):
        r = (r * x + a) % P
    return r


def poly_mul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % P
    return out


def poly_from_roots(roots):
    c = [1]
    for r in roots:
        c = poly_mul(c, [(-r) % P, 1])
    return c


def lagrange_eval(xs, ys, x):
    """O(n^2) interpolate-and-evaluate; the slow oracle of src/ec_fft.rs:883-907."""
    total = 0
    for i, (xi, yi) in enumerate(zip(xs, ys)):
        num, den = 1, 1
        for j, xj in enumerate(xs):
            if i != j:
                num = num * (x - xj) % P
                den = den * (xi - xj) % P
        total = (total + yi * num % P * fr_inv(den)) % P
    return total


def barycentric_eval(dom, bar_w, z_at_alpha, evals, alpha):
    """evaluate_poly_at_alpha_using_barycentric_weights, src/ec_fft.rs:455-491."""
    inv = fr_batch_inverse([(alpha - s) % P for s in dom])
    acc = 0
    for y, w, d in zip(evals, bar_w, inv):
        acc = (acc + y * w % P * d) % P
    return z_at_alpha * acc % P


# --------------------------------------------------------------------------------------------
# BLAKE3 (hash mode only) -- for the Fiat-Shamir transcript, src/proving.rs:79-198
# --------------------------------------------------------------------------------------------
_B3_IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_B3_PERM = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
_CHUNK_START, _CHUNK_END, _PARENT, _ROOT = 1, 2, 4, 8
_M32 = 0xFFFFFFFF


def _b3_g(s, a, b, c, d, mx, my):
    s[a] = (s[a] + s[b] + mx) & _M32
    s[d] = ((s[d] ^ s[a]) >> 16 | (s[d] ^ s[a]) << 16) & _M32
    s[c] = (s[c] + s[d]) & _M32
    s[b] = ((s[b] ^ s[c]) >> 12 | (s[b] ^ s[c]) << 20) & _M32
    s[a] = (s[a] + s[b] + my) & _M32
    s[d] = ((s[d] ^ s[a]) >> 8 | (s[d] ^ s[a]) << 24) & _M32
    s[c] = (s[c] + s[d]) & _M32
    s[b] = ((s[b] ^ s[c]) >> 7 | (s[b] ^ s[c]) << 25) & _M32


def _b3_compress(cv, block_words, counter, block_len, flags):
    s = list(cv) + _B3_IV[:4] + [counter & _M32, (counter >> 32) & _M32, block_len, flags]
    m = list(block_words)
    for r in range(7):
        _b3_g(s, 0, 4, 8, 12, m[0], m[1])
        _b3_g(s, 1, 5, 9, 13, m[2], m[3])
        _b3_g(s, 2, 6, 10, 14, m[4], m[5])
        _b3_g(s, 3, 7, 11, 15, m[6], m[7])
        _b3_g(s, 0, 5, 10, 15, m[8], m[9])
        _b3_g(s, 1, 6, 11, 12, m[10], m[11])
        _b3_g(s, 2, 7, 8, 13, m[12], m[13])
        _b3_g(s, 3, 4, 9, 14, m[14], m[15])
        if r < 6:
            m = [m[i] for i in _B3_PERM]
    return [s[i] ^ s[i + 8] for i in range(8)] + [s[i + 8] ^ cv[i] for i in range(8)]


def _b3_words(block: bytes):
    block = block + bytes(64 - len(block))
    return list(struct.unpack("<16I", block))


def _b3_chunk_output(chunk: bytes, counter: int):
    """Returns (cv_in, block_words, block_len, flags) of the chunk's last block."""
    cv = _B3_IV[:]
    blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
    for i, blk in enumerate(blocks):
        flags = (_CHUNK_START if i == 0 else 0) | (_CHUNK_END if i == len(blocks) - 1 else 0)
        if i == len(blocks) - 1:
            return cv, _b3_words(blk), len(blk), flags
        cv = _b3_compress(cv, _b3_words(blk), counter, 64, flags)[:8]


def blake3(data: bytes) -> bytes:
    chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)] or [b""]
    if len(chunks) == 1:
        cv, bw, bl, fl = _b3_chunk_output(chunks[0], 0)
        out = _b3_compress(cv, bw, 0, bl, fl | _ROOT)
        return struct.pack("<8I", *out[:8])
    # subtree stack per the BLAKE3 spec
    stack = []
    total = len(chunks)
    for idx, ch in enumerate(chunks[:-1]):
        cv, bw, bl, fl = _b3_chunk_output(ch, idx)
        new_cv = _b3_compress(cv, bw, idx, bl, fl)[:8]
        t = idx + 1
        while t & 1 == 0:
            left = stack.pop()
            new_cv = _b3_compress(_B3_IV, left + new_cv, 0, 64, _PARENT)[:8]
            t >>= 1
        stack.append(new_cv)
    cv, bw, bl, fl = _b3_chunk_output(chunks[-1], total - 1)
    # last chunk output, then fold the stack as parents; the final one is ROOT
    out_cv, out_bw, out_cnt, out_bl, out_fl = cv, bw, total - 1, bl, fl
    while stack:
        right = _b3_compress(out_cv, out_bw, out_cnt, out_bl, out_fl)[:8]
        left = stack.pop()
        out_cv, out_bw, out_cnt, out_bl, out_fl = _B3_IV[:], left + right, 0, 64, _PARENT
    out = _b3_compress(out_cv, out_bw, out_cnt, out_bl, out_fl | _ROOT)
    return struct.pack("<8I", *out[:8])


# --------------------------------------------------------------------------------------------
# Transcript (src/proving.rs:79-198)
# --------------------------------------------------------------------------------------------
def transcript_challenge(commit_p_bytes: bytes, public_inputs) -> int:
    srs_hash = blake3(b"")  # src/proving.rs:82-107 (buffer left empty)
    circuit_hash = blake3(b"")  # src/proving.rs:111-134 (buffers left empty)
    wc_hash = blake3(commit_p_bytes)  # :137-146
    buf = b"".join((x % P).to_bytes(29, "little") for x in public_inputs)  # :149-161
    pi_hash = blake3(buf)
    compile_hash = blake3(srs_hash + circuit_hash)
    runtime_hash = blake3(wc_hash + pi_hash)
    root = bytearray(blake3(compile_hash + runtime_hash))
    root[28:] = b"\0\0\0\0"  # :190
    return int.from_bytes(root, "little") % P


# --------------------------------------------------------------------------------------------
# R1CS (src/gnark_r1cs.rs) -- rows are ([(wire,coeff_id)...], [...], [...])
# --------------------------------------------------------------------------------------------
def eval_row(terms, coeffs, w):
    """R1CSInstance::eval_row, src/gnark_r1cs.rs:273-280."""
    acc = 0
    for wire, cid in terms:
        acc = (acc + coeffs[cid] * w[wire]) % P
    return acc


def evaluate_monomial_basis_poly(pub, alpha):
    """src/gnark_r1cs.rs:391-399."""
    acc, pw = 0, 1
    for x in pub:
        acc = (acc + x * pw) % P
        pw = pw * alpha % P
    return acc


def r1cs_with_vandermonde(rows, coeffs, dom, num_pub):
    """from_dump + update_to_include_vandermode_matrix_d, src/gnark_r1cs.rs:282-386.
    Returns (rows', coeffs', m)."""
    m = 1
    while m < len(rows):
        m *= 2
    rows = [(list(l), list(r), list(o)) for (l, r, o) in rows]
    coeffs = list(coeffs)
    minus_one = P - 1
    if minus_one in coeffs:
        idx1 = coeffs.index(minus_one)
    else:
        coeffs.append(minus_one)
        idx1 = len(coeffs) - 1
    assert len(dom) == m
    for i in range(m):
        if i >= len(rows):
            rows.append(([], [], []))
        pw = dom[i]
        for j in range(num_pub):
            wire = 1 + j
            if j == 0:
                rows[i][2].append((wire, idx1))
            else:
                coeffs.append((-pw) % P)
                rows[i][2].append((wire, len(coeffs) - 1))
                pw = pw * dom[i] % P
    return rows, coeffs, m


TOY_COEFFS = [1, 2]
TOY_ROWS = [  # src/dvsnark_test.rs:84-115; wires 1=0,o=1,w=2,y=3,z=4,x=5,t=6,s=7
    ([(5, 0)], [(5, 0)], [(3, 0)]),
    ([(3, 0), (4, 0)], [(0, 0)], [(2, 0)]),
    ([(4, 1)], [(0, 0)], [(6, 0)]),
    ([(5, 0), (6, 0)], [(0, 0)], [(7, 0)]),
    ([(2, 0), (7, 0)], [(0, 0)], [(1, 0)]),
]
TOY_PUBLIC = [24, 13]
TOY_PRIVATE = [9, 4, 3, 8, 11]


# --------------------------------------------------------------------------------------------
# Setup (src/srs.rs:112-167,177-361) and prove (src/proving.rs:426-688), brute-force flavoured:
# everything that the reference derives through vanish/exit/enter is computed here from the
# definition (products over the roots), which is only viable for small m.
# --------------------------------------------------------------------------------------------
def domain_tables(tree: FFTree):
    """Returns dict with D, D', z_poly (coeffs of Z_D), bar_wts, z_vals2inv, and the D' mirrors."""
    d, d2 = tree.both_domains()
    m = len(d)

    def deriv_at(dom, i):
        r = 1
        for j, x in enumerate(dom):
            if j != i:
                r = r * (dom[i] - x) % P
        return r

    z_poly = poly_from_roots(d)
    z_polyd = poly_from_roots(d2)
    return dict(
        D=d, D2=d2, m=m, z_poly=z_poly, z_polyd=z_polyd,
        bar_wts=[fr_inv(deriv_at(d, i)) for i in range(m)],
        bar_wtsd=[fr_inv(deriv_at(d2, i)) for i in range(m)],
        z_vals2inv=[fr_inv(poly_eval(z_poly, x)) for x in d2],  # 1/Z_D on D'
        z_vals2dinv=[fr_inv(poly_eval(z_polyd, x)) for x in d],  # 1/Z_D' on D
    )


def lagrange_at_tau(dom, z_poly, bar_w, tau):
    """compute_lagrange_basis_at_tau, src/ec_fft.rs:340-390."""
    z_tau = poly_eval(z_poly, tau)
    inv = fr_batch_inverse([(tau - s) % P for s in dom])
    return [z_tau * inv[i] % P * bar_w[i] % P for i in range(len(dom))]


def setup_srs_scalars(tree: FFTree, rows, coeffs, num_pub, trapdoor):
    """verifier_runs_setup + compute_srs_matrices, returning the *scalars* k such that each SRS
    base is k*G (src/srs.rs:126-160)."""
    tau, delta, eps = trapdoor
    tb = domain_tables(tree)
    m = tb["m"]
    rows2, coeffs2, m2 = r1cs_with_vandermonde(rows, coeffs, tb["D"], num_pub)
    assert m2 == m
    l_tau = lagrange_at_tau(tb["D"], tb["z_poly"], tb["bar_wts"], tau)
    l_taud = lagrange_at_tau(tb["D2"], tb["z_polyd"], tb["bar_wtsd"], tau)
    z_tau = poly_eval(tb["z_poly"], tau)
    zd_tau = poly_eval(tb["z_polyd"], tau)
    l_taul = [0] * (2 * m)  # src/ec_fft.rs:424-450
    for i in range(m):
        l_taul[2 * i] = l_tau[i] * zd_tau % P * tb["z_vals2dinv"][i] % P
        l_taul[2 * i + 1] = l_taud[i] * z_tau % P * tb["z_vals2inv"][i] % P
    # accumulate_m_values, src/srs.rs:53-84
    n_w = 1 + max(w for r in rows2 for part in r for (w, _) in part)
    m_vals = [0] * n_w
    d2 = delta * delta % P
    for i, (l, r, o) in enumerate(rows2):
        lt = l_tau[i]
        for w, c in l:
            m_vals[w] = (m_vals[w] + coeffs2[c] * lt) % P
        for w, c in r:
            m_vals[w] = (m_vals[w] + coeffs2[c] * lt % P * delta) % P
        for w, c in o:
            m_vals[w] = (m_vals[w] + coeffs2[c] * lt % P * d2) % P
    g_m = [v * eps % P for v in m_vals]
    g_q = [z_tau * d2 % P * l_taud[i] % P * eps % P for i in range(m)]
    g_k = [
        [l_tau[i] for i in range(m)],
        [l_tau[i] * delta % P for i in range(m)],
        [l_taul[i] * d2 % P for i in range(2 * m)],
    ]
    return dict(g_m=g_m, g_q=g_q, g_k=g_k, tables=tb, rows=rows2, coeffs=coeffs2)


def prove_scalars(tree: FFTree, setup, public_inputs, private_inputs, alpha_fn):
    """
    Proof::prove (src/proving.rs:426-688) stage by stage.  The MSMs are kept symbolic as
    (scalars, base-scalars) so callers can evaluate them with any MSM implementation;
    alpha_fn(commit_p_scalar) -> alpha supplies the Fiat-Shamir challenge.
    Returns dict with every intermediate the GPU path reproduces.
    """
    tb = setup["tables"]
    rows, coeffs = setup["rows"], setup["coeffs"]
    m = tb["m"]
    d, d2 = tb["D"], tb["D2"]
    w = [1] + list(public_inputs) + list(private_inputs)  # :355-359
    a = [eval_row(r[0], coeffs, w) for r in rows]
    b = [eval_row(r[1], coeffs, w) for r in rows]
    c = [eval_row(r[2], coeffs, w) for r in rows]
    iv = [evaluate_monomial_basis_poly(public_inputs, x) for x in d]  # :369-376
    for i in range(m):
        assert a[i] * b[i] % P == (c[i] + iv[i]) % P, f"constraint {i}"  # :389-395
    a2, b2, c2, i2 = (tree.extend(v) for v in (a, b, c, iv))  # :410-422
    r2 = [(a2[i] * b2[i] - i2[i]) % P for i in range(m)]  # :492-495
    q2 = [(r2[i] - c2[i]) * tb["z_vals2inv"][i] % P for i in range(m)]  # :505-508
    # commitments as discrete logs wrt G: msm_gm = <w, g_m>, msm_q = <q2, g_q>
    dl_gm = sum(x * y for x, y in zip(w, setup["g_m"])) % P
    dl_q = sum(x * y for x, y in zip(q2, setup["g_q"])) % P
    dl_commit_p = (dl_gm + dl_q) % P
    alpha = alpha_fn(dl_commit_p)
    assert alpha not in d and alpha not in d2
    z_alpha = poly_eval(tb["z_poly"], alpha)
    a0 = barycentric_eval(d, tb["bar_wts"], z_alpha, a, alpha)
    b0 = barycentric_eval(d, tb["bar_wts"], z_alpha, b, alpha)
    i0 = barycentric_eval(d, tb["bar_wts"], z_alpha, iv, alpha)
    r0 = (a0 * b0 - i0) % P
    dinv = fr_batch_inverse([(x - alpha) % P for x in d])
    dinv2 = fr_batch_inverse([(x - alpha) % P for x in d2])
    ka = [(a[i] - a0) * dinv[i] % P for i in range(m)]
    kb = [(b[i] - b0) * dinv[i] % P for i in range(m)]
    r = [(a[i] * b[i] - iv[i]) % P for i in range(m)]
    kr = []
    for i in range(m):  # :644-654 interleaved [D_i, D'_i]
        kr.append((r[i] - r0) * dinv[i] % P)
        kr.append((r2[i] - r0) * dinv2[i] % P)
    s_k = ka + kb + kr  # :674-677
    g_k = setup["g_k"][0] + setup["g_k"][1] + setup["g_k"][2]  # :666-672
    dl_kzg = sum(x * y for x, y in zip(s_k, g_k)) % P
    return dict(a=a, b=b, c=c, i=iv, a2=a2, b2=b2, c2=c2, i2=i2, r2=r2, q2=q2, w=w,
                dl_gm=dl_gm, dl_q=dl_q, dl_commit_p=dl_commit_p, alpha=alpha, z_alpha=z_alpha,
                a0=a0, b0=b0, i0=i0, r0=r0, ka=ka, kb=kb, kr=kr, s_k=s_k, dl_kzg=dl_kzg)


def verify_dl(trapdoor, public_inputs, dl_commit_p, dl_kzg, a0, b0, alpha) -> bool:
    """SRS::verify (src/srs.rs:374-428) on discrete logs: v0*K + u0*G == P."""
    tau, delta, eps = trapdoor
    i0 = evaluate_monomial_basis_poly(public_inputs, alpha)
    r0 = (a0 * b0 - i0) % P
    u0 = (a0 + delta * b0 + delta * delta % P * r0) % P * eps % P
    v0 = (tau - alpha) * eps % P
    return (v0 * dl_kzg + u0) % P == dl_commit_p % P


def proof_to_bits(commit_p: bytes, kzg_k: bytes, a0: int, b0: int):
    """Proof::to_bits, src/proving.rs:691-718."""
    bits = []
    for blob in (commit_p, kzg_k):
        for byte in blob:
            bits += [(byte >> i) & 1 for i in range(8)]
    bits += frbits_from_fr(a0) + frbits_from_fr(b0)
    return bits


# --------------------------------------------------------------------------------------------
# deterministic test-vector RNG shared with the C/HIP side
# --------------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        """uniform in [0,p) by rejection on 232 bits"""
        while True:
            v = self.next() | (self.next() << 64) | (self.next() << 128) | ((self.next() & ((1 << 40) - 1)) << 192)
            if v < P:
                return v
