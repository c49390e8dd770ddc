"""A/B of library builds on ONE box: python tools/ab_libs.py ab/libdvpari_a.so ab/libdvpari_b.so ...  (each in its own child process,
interleaved ROUNDS times so that box drift does not masquerade as a build effect).  Child: 2^20 proof, 8 timed proofs, per-stage HIP events."""
import ctypes as C, importlib, json, os, statistics, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
if os.environ.get("AB_CHILD"):
    import numpy as np, torch
    dvp = importlib.import_module("dv-pari_amd")
    log_m = int(os.environ.get("LOG_M", "20"))
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    for _ in range(3):
        p = pv.prove_dev(w.data_ptr(), 0)
    out = []
    for rep in range(int(os.environ.get("AB_REPS", "3"))):
        dvp.lib.dvp_profile_reset(); dvp.lib.dvp_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            p = pv.prove_dev(w.data_ptr(), 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8 * 1e3
        dvp.lib.dvp_profile_enable(0)
        st = {}
        for name in ("msm_total", "msm_sort", "msm_affine_round0", "msm_affine_rest", "msm_tail", "extend_total"):
            ms, n = C.c_double(0), C.c_uint64(0)
            dvp.check(dvp.lib.dvp_profile_read(name.encode(), C.byref(ms), C.byref(n)))
            st[name] = round(ms.value / 8, 3)
        out.append({"ms": round(dt, 3), **st})
    print("AB_RESULT " + json.dumps({"proof": p.to_bytes().hex(), "runs": out}), flush=True)
    sys.exit(0)
libs = sys.argv[1:]
rounds = int(os.environ.get("ROUNDS", "2"))
res = {l: [] for l in libs}
proofs = set()
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, AB_CHILD="1", DVP_LIB=os.path.abspath(l))
        o = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [x for x in o.stdout.splitlines() if x.startswith("AB_RESULT ")]
        if not line:
            print(l, "FAILED", o.stderr[-2000:], flush=True)
            continue
        d = json.loads(line[0][10:])
        proofs.add(d["proof"])
        res[l] += d["runs"]
        print(l, "round", r, [x["ms"] for x in d["runs"]], flush=True)
for l in libs:
    if not res[l]:
        continue
    keys = res[l][0].keys()
    print(os.path.basename(l), {k: round(statistics.median(x[k] for x in res[l]), 3) for k in keys}, flush=True)
print("proof bytes identical across builds:", len(proofs) == 1)
