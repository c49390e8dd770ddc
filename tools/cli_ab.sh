#!/bin/bash
# Runs ON THE MI355X BOX: throughput of the compiled C++ host (examples/dvp_prove_cli.cpp) through dvp_prove_cache_dir from one and two
# host threads, for the in-tree library and for every ab/libdvpari_*.so (each gets an executable of its own, linked against a private copy:
# never two builds of the library in one process -- an LD_PRELOAD'ed second copy registers its kernels beside the first and faults).
# usage: gpurun -- 'bash tools/cli_ab.sh [repeat=40]'
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
REPEAT=${1:-40} python3 - <<'PY'
import glob, importlib, os, sys, subprocess, tempfile, shutil
R = os.getcwd(); sys.path.insert(0, R)
dvp = importlib.import_module("dv-pari_amd")
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
tmp = tempfile.mkdtemp(prefix="dvp_cli_", dir=os.path.join(R, "gpurun_out"))
try:
    def build_exe(tag, so):
        d = os.path.join(tmp, "lib_" + tag); os.mkdir(d)
        shutil.copy(so, os.path.join(d, "libdvpari_hip.so"))
        exe = os.path.join(tmp, "cli_" + tag)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(R, "include"), os.path.join(R, "examples", "dvp_prove_cli.cpp"),
                               "-L" + d, "-ldvpari_hip", "-Wl,-rpath," + d, "-pthread", "-o", exe])
        return exe
    A, g = dvp.artifacts, dvp.gnark_r1cs
    inst, pub, prv = g.synthetic_dense(20)
    cache = os.path.join(tmp, "cache"); os.mkdir(cache)
    inst.write_dump_file(os.path.join(cache, A.R1CS_CONSTRAINTS_FILE))
    g.write_witness_to_file(os.path.join(cache, A.R1CS_WITNESS_FILE), [1] + pub + prv)
    td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    _, pv = dvp.srs.verifier_runs_setup_cache_dir(td, cache, len(pub), write_precomputes=False); pv.close()
    libs = [("in-tree", build_exe("intree", os.path.join(R, "dv-pari_amd", "libdvpari_hip.so")))]
    libs += [(os.path.basename(p), build_exe(os.path.basename(p)[:-3], p)) for p in sorted(glob.glob(os.path.join(R, "ab", "libdvpari_*.so")))]
    for rep in range(2):
        for name, exe in libs:
            for threads in (1, 2):
                env = dict(os.environ, DVP_NO_TORCH_PRELOAD="1")
                out = subprocess.run([exe, cache, str(len(pub)), "--repeat", os.environ.get("REPEAT", "40"), "--threads", str(threads)],
                                     capture_output=True, text=True, env=env, timeout=600)
                print(name, (out.stderr.strip().splitlines() or ["?"])[-1], flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
PY
