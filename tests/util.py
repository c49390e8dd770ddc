import numpy as np

import pyref as o

MASK64 = (1 << 64) - 1


def to_limbs(vals, words=4):
    return np.array([[(int(v) >> (64 * i)) & MASK64 for i in range(words)] for v in vals], dtype=np.uint64).reshape(len(vals), words)


def from_limbs(arr):
    a = np.ascontiguousarray(arr, dtype="<u8")
    a = a.reshape(-1, a.shape[-1])
    w = a.shape[1] * 8
    raw = a.tobytes()
    return [int.from_bytes(raw[i * w:(i + 1) * w], "little") for i in range(a.shape[0])]


def pts_to_np(pts):
    """list of (x,y) -> [n,8] uint64"""
    return to_limbs([c for p in pts for c in p]).reshape(len(pts), 8)


def np_to_pt(xy, inf=False):
    if inf:
        return None
    v = from_limbs(np.asarray(xy).reshape(2, 4))
    return (v[0], v[1])


def rand_fr_np(n, seed):
    """n uniform-ish canonical scalars (< 2^231 < p) as [n,4] uint64, fast"""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 39) - 1)
    return s


def np_dot_mod(a, b):
    """sum a_i*b_i mod p for two [n,4] limb arrays (python ints)"""
    return sum(x * y for x, y in zip(from_limbs(a), from_limbs(b))) % o.P


def np_dot_mod_fast(a, b):
    """sum a_i*b_i mod p for two [n,4] limb arrays, EXACT, in numpy only (no product code): the operands are cut
    into 16-bit limbs and the 16 x 16 limb-pair sums come from float64 matrix products over chunks of 2^18 rows
    (every partial sum < 2^32 * 2^18 = 2^50 is exact in a double); the chunks are recombined with python ints.
    ~1 s per 10 M terms, against ~1 us per term for a python loop -- this is what lets the full-size tests apply the
    discrete-log identity sum s_i (k_i G) = (sum s_i k_i) G."""
    A = np.ascontiguousarray(a, dtype="<u8").reshape(-1, 4).view("<u2").reshape(-1, 16)
    B = np.ascontiguousarray(b, dtype="<u8").reshape(-1, 4).view("<u2").reshape(-1, 16)
    assert A.shape == B.shape
    total = 0
    step = 1 << 18
    for lo in range(0, A.shape[0], step):
        M = A[lo:lo + step].astype(np.float64).T @ B[lo:lo + step].astype(np.float64)
        for i in range(16):
            for j in range(16):
                total += int(M[i, j]) << (16 * (i + j))
    return total % o.P


# ---- adversarial scalars for the tau-adic recoding (dv-pari_amd/csrc/tau.cuh) ----------------------------------------
# delta = D0 + D1 tau with N(delta) = p; tau acts on E[r] as lambda = -D0/D1 mod p
TAU_D0 = 0x000325402DCB0ED1DA32C0F4BA75BB3B
TAU_D1 = 0x000882D72D7AE36E16AA143CCB36BEE6


def tau_adversarial_scalars(seed=7, n_boundary=40):
    """scalars whose partial reduction modulo delta lands at the corners of the fundamental region (largest norm,
    hence the longest {0,1}-digit expansions), and scalars that sit on the rounding boundaries of the two fixed-point
    quotients round(s * A_i / 2^256) of tau_partial_reduce"""
    import random

    p = o.P
    assert TAU_D0 * TAU_D0 - TAU_D0 * TAU_D1 + 2 * TAU_D1 * TAU_D1 == p
    lam = (-TAU_D0 * pow(TAU_D1, -1, p)) % p
    assert (lam * lam + lam + 2) % p == 0
    out = []
    for e0, e1 in ((1, -1), (-1, 1), (1, 1), (-1, -1), (1, 0), (0, 1), (-1, 0), (0, -1)):
        n0 = e0 * TAU_D0 - 2 * e1 * TAU_D1          # (e0/2 + e1/2 tau) * delta, doubled
        n1 = e0 * TAU_D1 + e1 * TAU_D0 - e1 * TAU_D1
        for a in range(-3, 4):
            for b in range(-3, 4):
                out.append(((n0 // 2 + a) + (n1 // 2 + b) * lam) % p)
    rnd = random.Random(seed)
    a0 = ((TAU_D1 - TAU_D0) << 256) // p
    a1 = (TAU_D1 << 256) // p
    for _ in range(n_boundary):
        k = rnd.randrange(1 << 116)
        for A in (a0, a1):
            s0 = ((2 * k + 1) << 255) // A
            out += [(s0 + d) % p for d in (-1, 0, 1)]
    return out
