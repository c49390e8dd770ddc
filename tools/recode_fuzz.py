"""One-off fuzz of k_recode_slide (through dvp_debug_recode_slide): for every window size 8..21 and N random + structured
scalars, the entry words must be disjoint odd windows of <= c digits whose value sum d_j lambda^j equals the scalar mod r,
within the slot bound.  python tools/recode_fuzz.py [N per window size]"""
import ctypes as C, importlib, os, random, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
import pyref as o
import c_oracle as co
from util import to_limbs, tau_adversarial_scalars, slide_slots, TAU_D0, TAU_D1
dvp = importlib.import_module("dv-pari_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lam = (-TAU_D0 * pow(TAU_D1, -1, o.P)) % o.P
probe = list(co.tau_digits(12345))
if sum(d * pow(lam, j, o.P) for j, d in enumerate(probe)) % o.P != 12345:
    lam = (-1 - lam) % o.P
pw = [pow(lam, j, o.P) for j in range(262)]
rnd = random.Random(2027)
structured = [0, 1, 2, 3, o.P - 1, o.P - 2] + [(1 << k) % o.P for k in range(0, 232, 3)] + [((1 << k) - 1) % o.P for k in range(1, 232, 5)]
structured += [int("55" * 29, 16) % o.P, int("aa" * 29, 16) % o.P, int("0f" * 29, 16) % o.P] + tau_adversarial_scalars()
bad = 0
for c in range(8, 22):
    vals = structured + [rnd.randrange(o.P) for _ in range(N)] + [rnd.randrange(1 << rnd.randrange(1, 232)) for _ in range(N // 4)]
    s = to_limbs(vals)
    slots = C.c_int(0)
    dvp.check(dvp.lib.dvp_debug_recode_slide(None, 0, c, None, C.byref(slots)), "slots")
    assert slots.value == slide_slots(c)
    words = np.zeros((slots.value, len(vals)), dtype=np.uint32)
    dvp.check(dvp.lib.dvp_debug_recode_slide(s.ctypes.data, len(vals), c, words.ctypes.data, C.byref(slots)), "recode")
    pos = ((words >> 20) & 0xFF).astype(np.int64)
    pat = (2 * (words & 0xFFFFF).astype(np.int64) + 1) * (words != 0)
    mx = 0
    for i, x in enumerate(vals):
        value, end = 0, 0
        for sl in range(slots.value):
            w = int(words[sl, i])
            if not w:
                assert not words[sl:, i].any()
                break
            p_, v = int(pos[sl, i]), int(pat[sl, i])
            assert p_ >= end and v < (1 << c) and (v & 1), (c, i, hex(x))
            end = p_ + v.bit_length()
            t = 0
            while v:
                if v & 1:
                    value += pw[p_ + t]
                v >>= 1
                t += 1
        if value % o.P != x % o.P or end > 240:
            bad += 1
            print("MISMATCH", c, hex(x))
        mx = max(mx, end)
    print(f"c={c}: {len(vals)} scalars ok, longest expansion {mx} digits, {slots.value} slots", flush=True)
print("fuzz done, mismatches:", bad)
sys.exit(1 if bad else 0)
