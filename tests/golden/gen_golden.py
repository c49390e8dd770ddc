"""
Generates tests/golden/oracle_vectors.json from oracle/pyref.py (the Python big-int restatement).
These are regression vectors for the C oracle and the HIP path: the THIRD-PARTY pin of the group
law is tests/golden/k233_openssl.json (OpenSSL), the pin of ECFFT is the O(n^2) Lagrange identity
(tests/test_oracle.py); this file freezes their outputs so every backend is compared on the same
bytes.  Run: python tests/golden/gen_golden.py
"""
import json, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
from pyref import *  # noqa

rng = SplitMix64(0x5EED0001)
out = {}
gfv = []
for _ in range(16):
    a = rng.next() | (rng.next() << 64) | (rng.next() << 128) | ((rng.next() & ((1 << 41) - 1)) << 192)
    b = rng.next() | (rng.next() << 64) | (rng.next() << 128) | ((rng.next() & ((1 << 41) - 1)) << 192)
    gfv.append({"a": hex(a), "b": hex(b), "mul": hex(gf_mul(a, b)), "sqr": hex(gf_sqr(a)), "inv": hex(gf_inv(a)),
                "sqrt": hex(gf_sqrt(a)), "trace": gf_trace(a)})
out["gf233"] = gfv

# ECFFT: 16-leaf tree (the size of the reference's own extend test, src/ec_fft.rs:883-907) and 64
ec = {}
for log_n in (4, 6):
    t = FFTree(log_n)
    n = 1 << log_n
    ev = [rng.fr() for _ in range(n // 2)]
    co = [rng.fr() for _ in range(n)]
    ec[str(log_n)] = {
        "leaves": [hex(x) for x in t.leaves()],
        "extend_in": [hex(x) for x in ev], "extend_out": [hex(x) for x in t.extend(ev)],
        "coeffs": [hex(x) for x in co], "enter_out": [hex(x) for x in t.enter(co)],
        "vanish_even_at_7": hex(t.vanish_even_at(7)),
    }
ts = FFTree(4, shifted=True, base_log_n=5)
ec["4_shifted_base5"] = {"leaves": [hex(x) for x in ts.leaves()]}
out["ecfft"] = ec

# xsk233 codec candidate (PARITY UNPINNED vs xs233) and a small MSM
cod = []
for k in (1, 2, 3, 0xDEADBEEF, P - 1):
    pt = k233_mul(k, G_STD)
    cod.append({"k": hex(k), "enc": xsk233_encode(pt).hex()})
out["xsk233_candidate"] = cod
ks = [rng.fr() for _ in range(6)]
ss = [rng.fr() for _ in range(6)]
pts = [k233_mul(k, G_STD) for k in ks]
res = k233_msm(ss, pts)
out["msm6"] = {"k": [hex(k) for k in ks], "s": [hex(s) for s in ss], "x": hex(res[0]), "y": hex(res[1])}

# BLAKE3 (official vectors for len 0 / 1 / 1025 with input byte i % 251) + transcript challenge
out["blake3"] = {str(n): blake3(bytes(i % 251 for i in range(n))).hex() for n in (0, 1, 64, 1024, 1025, 2049)}
out["challenge_toy"] = hex(transcript_challenge(bytes(range(30)), TOY_PUBLIC))

# toy R1CS (src/dvsnark_test.rs:131-180) with a fixed trapdoor
trap = (rng.fr(), rng.fr(), rng.fr())
tree = FFTree(4)
st = setup_srs_scalars(tree, TOY_ROWS, TOY_COEFFS, 2, trap)


def alpha_fn(dl):
    return transcript_challenge(xsk233_encode(k233_mul(dl, G_STD)), TOY_PUBLIC)


pr = prove_scalars(tree, st, TOY_PUBLIC, TOY_PRIVATE, alpha_fn)
assert verify_dl(trap, TOY_PUBLIC, pr["dl_commit_p"], pr["dl_kzg"], pr["a0"], pr["b0"], pr["alpha"])
toy = {"trapdoor": [hex(x) for x in trap]}
for key in ("a", "b", "c", "i", "a2", "b2", "c2", "i2", "q2", "ka", "kb", "kr"):
    toy[key] = [hex(x) for x in pr[key]]
for key in ("alpha", "a0", "b0", "i0", "r0", "dl_commit_p", "dl_kzg", "z_alpha"):
    toy[key] = hex(pr[key])
toy["g_m"] = [hex(x) for x in st["g_m"]]
toy["g_q"] = [hex(x) for x in st["g_q"]]
toy["g_k"] = [[hex(x) for x in v] for v in st["g_k"]]
toy["bar_wts"] = [hex(x) for x in st["tables"]["bar_wts"]]
toy["z_vals2inv"] = [hex(x) for x in st["tables"]["z_vals2inv"]]
toy["commit_p"] = xsk233_encode(k233_mul(pr["dl_commit_p"], G_STD)).hex()
toy["kzg_k"] = xsk233_encode(k233_mul(pr["dl_kzg"], G_STD)).hex()
out["toy"] = toy

with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
    json.dump(out, f, indent=0)
print("wrote oracle_vectors.json")
