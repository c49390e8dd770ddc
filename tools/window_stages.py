"""2^20 proof with the fixed-base window size forced (both MSMs): total and per-stage HIP-event times per proof -- what a window
more or less moves between the pair rounds and the merge tree.  python tools/window_stages.py 19 20 21"""
import ctypes as C, importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(20)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
srs = None
for c in [int(x) for x in sys.argv[1:]] or [0]:
    with dvp.tune(**({"DVP_MSM_FIXED_C": c} if c else {})):
        pv = dvp.proving.Prover(inst)
        if srs is None:
            srs = dvp.srs.verifier_runs_setup(pv, inst, td)
        pv.set_srs(srs)
        for _ in range(2):
            pv.prove_dev(w.data_ptr(), 0)
        dvp.lib.dvp_profile_reset(); dvp.lib.dvp_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            pv.prove_dev(w.data_ptr(), 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8 * 1e3
        dvp.lib.dvp_profile_enable(0)
        st = {}
        for name in ("msm_total", "msm_sort", "msm_affine_round0", "msm_affine_rest", "msm_tail"):
            ms, n = C.c_double(0), C.c_uint64(0)
            dvp.check(dvp.lib.dvp_profile_read(name.encode(), C.byref(ms), C.byref(n)))
            st[name] = ms.value / 8
        other = st["msm_total"] - sum(v for k, v in st.items() if k != "msm_total")
        print(f"c={c or 'default'} plans {pv.msm_plan(0)} {pv.msm_plan(1)}: proof {dt:.2f} ms | sort {st['msm_sort']:.2f} round0 {st['msm_affine_round0']:.2f} "
              f"later rounds {st['msm_affine_rest']:.2f} merge+tail {st['msm_tail']:.2f} reducer+bookkeeping {other:.2f}", flush=True)
        pv.close()
