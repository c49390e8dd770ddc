"""The compiled C++ host (examples/dvp_prove_cli.cpp, public header only) over a 2^log_m cache_dir written here: proofs per second
through the reference's own signature Proof::prove(cache_dir, public, private) = dvp_prove_cache_dir, from 1 and 2 host threads.
python tools/cli_throughput.py [log_m]"""
import importlib, os, shutil, subprocess, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
dvp = importlib.import_module("dv-pari_amd")
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tmp = tempfile.mkdtemp(prefix="dvp_cli_", dir=os.path.join(R, "gpurun_out") if os.path.isdir(os.path.join(R, "gpurun_out")) else None)
try:
    exe = os.path.join(tmp, "dvp_prove_cli")
    libdir = os.path.join(R, "dv-pari_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(R, "include"), os.path.join(R, "examples", "dvp_prove_cli.cpp"),
                           "-L" + libdir, "-ldvpari_hip", "-Wl,-rpath," + libdir, "-pthread", "-o", exe])
    A, g = dvp.artifacts, dvp.gnark_r1cs
    t0 = time.time()
    inst, pub, prv = g.synthetic_dense(log_m)
    cache = os.path.join(tmp, "cache"); os.mkdir(cache)
    inst.write_dump_file(os.path.join(cache, A.R1CS_CONSTRAINTS_FILE))
    g.write_witness_to_file(os.path.join(cache, A.R1CS_WITNESS_FILE), [1] + pub + prv)
    td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    _, pv = dvp.srs.verifier_runs_setup_cache_dir(td, cache, len(pub), write_precomputes=False)
    pv.close()
    print(f"cache_dir for 2^{log_m} written in {time.time() - t0:.0f}s", flush=True)
    env = dict(os.environ, DVP_NO_TORCH_PRELOAD="1")
    for threads in (1, 2):
        out = subprocess.run([exe, cache, str(len(pub)), "--repeat", "20", "--threads", str(threads)], capture_output=True, text=True, env=env, timeout=900)
        assert out.returncode == 0, out.stderr
        print(out.stderr.strip().splitlines()[-1], flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
