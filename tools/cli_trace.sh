#!/bin/bash
# Runs ON THE MI355X BOX: rocprofv3 --kernel-trace of the compiled C++ host proving through dvp_prove_cache_dir from two host threads
# (where do the two provers' kernels overlap, where do they wait?).  Output: gpurun_out/cli_trace/
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
OUT=$R/gpurun_out/cli_trace; rm -rf $OUT; mkdir -p $OUT
python3 - <<'PY'
import importlib, os, sys, subprocess
R = os.getcwd(); sys.path.insert(0, R)
dvp = importlib.import_module("dv-pari_amd")
out = os.path.join(R, "gpurun_out", "cli_trace")
libdir = os.path.join(R, "dv-pari_amd")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(R, "include"), os.path.join(R, "examples", "dvp_prove_cli.cpp"),
                       "-L" + libdir, "-ldvpari_hip", "-Wl,-rpath," + libdir, "-pthread", "-o", os.path.join(out, "cli")])
A, g = dvp.artifacts, dvp.gnark_r1cs
inst, pub, prv = g.synthetic_dense(20)
cache = os.path.join(out, "cache"); os.mkdir(cache)
inst.write_dump_file(os.path.join(cache, A.R1CS_CONSTRAINTS_FILE))
g.write_witness_to_file(os.path.join(cache, A.R1CS_WITNESS_FILE), [1] + pub + prv)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
_, pv = dvp.srs.verifier_runs_setup_cache_dir(td, cache, len(pub), write_precomputes=False); pv.close()
open(os.path.join(out, "npub"), "w").write(str(len(pub)))
PY
cd /tmp && export TMPDIR=/tmp DVP_NO_TORCH_PRELOAD=1
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/prof -o t --output-format csv -- $OUT/cli $OUT/cache $(cat $OUT/npub) --repeat 8 --threads 2 > $OUT/run.log 2>&1 || true
tail -n 2 $OUT/run.log
rm -rf $OUT/cache
