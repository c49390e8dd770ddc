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
