cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|power" | head -5
python3 - <<'PY' &
import importlib, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
n = 1 << 22
rng = np.random.default_rng(1)
def rs(seed):
    r = np.random.default_rng(seed)
    s = r.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * 2 + r.integers(0, 2, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 39) - 1)
    return s
k, s = rs(1), rs(2)
xy, inf = dvp.curve.point_scalar_mul_gen_batch(k)
fb = dvp.curve.FixedBaseMsm(xy)
print("table", fb.table(), flush=True)
t0 = time.time(); cnt = 0
while time.time() - t0 < 12:
    fb.run(s); cnt += 1
print("msm loop: %.2f ms per 2^22-point fixed-base MSM (incl. 128 MB H2D of scalars)" % ((time.time() - t0) / cnt * 1e3))
PY
P=$!
sleep 9
for i in $(seq 1 25); do rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import json,sys
c=json.load(sys.stdin)['card0']
print(c.get('Current Socket Graphics Package Power (W)'), c.get('sclk clock speed:'))"; sleep 0.3; done
wait $P
