"""
Generates tests/golden/k233_openssl.json: sect233k1 scalar multiples k*G computed by OpenSSL's
libcrypto (an independent third-party implementation) -- the known-answer pin for the oracle's
K-233 group law.  Run in the build container (libcrypto.so.3 present); the JSON is committed.
"""
import ctypes, ctypes.util, json, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
from pyref import SplitMix64, P  # noqa: E402

lib = ctypes.CDLL(ctypes.util.find_library("crypto"))
vp = ctypes.c_void_p
for name, res, args in [
    ("EC_GROUP_new_by_curve_name", vp, [ctypes.c_int]),
    ("EC_POINT_new", vp, [vp]),
    ("BN_new", vp, []),
    ("BN_CTX_new", vp, []),
    ("BN_hex2bn", ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_char_p]),
    ("BN_bn2hex", vp, [vp]),
    ("EC_POINT_mul", ctypes.c_int, [vp, vp, vp, vp, vp, vp]),
    ("EC_POINT_add", ctypes.c_int, [vp, vp, vp, vp, vp]),
    ("EC_POINT_get_affine_coordinates", ctypes.c_int, [vp, vp, vp, vp, vp]),
    ("EC_POINT_is_at_infinity", ctypes.c_int, [vp, vp]),
]:
    f = getattr(lib, name)
    f.restype, f.argtypes = res, args

NID_sect233k1 = 726
grp = lib.EC_GROUP_new_by_curve_name(NID_sect233k1)
assert grp
ctx = lib.BN_CTX_new()


def bn(v):
    b = vp()
    assert lib.BN_hex2bn(ctypes.byref(b), b"%x" % v)
    return b


def hexof(b):
    return int(ctypes.cast(lib.BN_bn2hex(b), ctypes.c_char_p).value, 16)


def mulgen(k):
    pt = lib.EC_POINT_new(grp)
    assert lib.EC_POINT_mul(grp, pt, bn(k), None, None, ctx)
    if lib.EC_POINT_is_at_infinity(grp, pt):
        return None
    x, y = lib.BN_new(), lib.BN_new()
    assert lib.EC_POINT_get_affine_coordinates(grp, pt, x, y, ctx)
    return [hexof(x), hexof(y)]


rng = SplitMix64(0x5EED0000)
ks = [1, 2, 3, 4, 5, 7, 8, 15, 16, 255, 256, 65535, 65537, P - 1, P - 2, (P + 1) // 2] + [rng.fr() for _ in range(48)]
out = {"curve": "sect233k1", "source": "OpenSSL libcrypto EC_POINT_mul", "vectors": []}
for k in ks:
    pt = mulgen(k)
    out["vectors"].append({"k": hex(k), "x": hex(pt[0]), "y": hex(pt[1])})
with open(os.path.join(os.path.dirname(__file__), "k233_openssl.json"), "w") as f:
    json.dump(out, f, indent=0)
print("wrote", len(ks), "vectors")
