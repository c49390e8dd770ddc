#!/bin/bash
# Runs ON THE MI355X BOX: samples rocm-smi power / clocks while (a) the multiplier microbenchmark and (b) the 2^20 prove loop
# run -- the evidence behind "the pair rounds are power-limited" (DESIGN.md section 9).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/power
rm -rf $OUT; mkdir -p $OUT
sample() {  # $1 = tag, $2 = pid to follow
  while kill -0 $2 2>/dev/null; do
    rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' >> $OUT/$1.jsonl; echo >> $OUT/$1.jsonl
    sleep 0.2
  done
}
cd $ROOT
python3 - <<'PY' &
import ctypes as C, importlib, sys, time
sys.path.insert(0, ".")
dvp = importlib.import_module("dv-pari_amd")
r = C.c_double(0)
t0 = time.time()
while time.time() - t0 < 6:
    dvp.check(dvp.lib.dvp_ubench_gf_mul(4000, C.byref(r)), "ubench")
print("ubench", r.value)
PY
P=$!; sleep 2.5; sample ubench $P; wait $P
N_PROOFS=500 LOG_M=20 python3 tools/host_gap.py > $OUT/prove.log 2>&1 &
P=$!; sleep 7; sample prove $P; wait $P; cat $OUT/prove.log | grep -v amdgpu
python3 - <<PY
import json
for tag in ("ubench", "prove"):
    rows = []
    for line in open("$OUT/%s.jsonl" % tag):
        line = line.strip()
        if not line: continue
        try: rows.append(json.loads(line))
        except Exception: pass
    print(tag, len(rows), "samples")
    for r in rows[:: max(1, len(rows) // 12)]:
        c = r.get("card0", {})
        print("   ", {k: v for k, v in c.items() if "ower" in k or "sclk" in k or "mclk" in k})
PY
