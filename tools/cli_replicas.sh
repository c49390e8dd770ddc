#!/bin/bash
# Runs ON THE MI355X BOX: the compiled C++ host through dvp_prove_cache_dir from one and two host threads with DVP_CACHE_REPLICAS = 1 / 2
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<'PY'
import importlib, os, sys, subprocess, tempfile, shutil
R = os.getcwd(); sys.path.insert(0, R)
dvp = importlib.import_module("dv-pari_amd")
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
tmp = tempfile.mkdtemp(prefix="dvp_cli_", dir=os.path.join(R, "gpurun_out"))
try:
    exe = os.path.join(tmp, "cli"); libdir = os.path.join(R, "dv-pari_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(R, "include"), os.path.join(R, "examples", "dvp_prove_cli.cpp"),
                           "-L" + libdir, "-ldvpari_hip", "-Wl,-rpath," + libdir, "-pthread", "-o", exe])
    A, g = dvp.artifacts, dvp.gnark_r1cs
    inst, pub, prv = g.synthetic_dense(20)
    cache = os.path.join(tmp, "cache"); os.mkdir(cache)
    inst.write_dump_file(os.path.join(cache, A.R1CS_CONSTRAINTS_FILE))
    g.write_witness_to_file(os.path.join(cache, A.R1CS_WITNESS_FILE), [1] + pub + prv)
    td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    _, pv = dvp.srs.verifier_runs_setup_cache_dir(td, cache, len(pub), write_precomputes=False); pv.close()
    for rep in range(2):
        for replicas in (1, 2):
            for threads in (1, 2, 3):
                env = dict(os.environ, DVP_NO_TORCH_PRELOAD="1", DVP_CACHE_REPLICAS=str(replicas))
                out = subprocess.run([exe, cache, str(len(pub)), "--repeat", "40", "--threads", str(threads)], capture_output=True, text=True, env=env, timeout=600)
                print("replicas", replicas, (out.stderr.strip().splitlines() or ["?"])[-1], flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
PY
