"""2^20 proof under each fixed-base table flavour: total, stages, table bytes -- python tools/flavours.py [log_m]"""
import ctypes as C, importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
srs, ref = None, None
FL = [("sliding, binary digits (default)", {}), ("sliding, tau-adic", {"DVP_MSM_SLIDE": 1}),
      ("aligned, signed binary digits", {"DVP_MSM_SLIDE": 0}), ("aligned, signed, c=18", {"DVP_MSM_SLIDE": 0, "DVP_MSM_FIXED_C": 18}),
      ("aligned, signed, c=17", {"DVP_MSM_SLIDE": 0, "DVP_MSM_FIXED_C": 17}),
      ("aligned, tau-adic", {"DVP_MSM_SLIDE": 0, "DVP_MSM_ALIGNED_SIGNED": 0})]
for name, knobs in FL:
    with dvp.tune(**knobs):
        pv = dvp.proving.Prover(inst)
        if srs is None:
            srs = dvp.srs.verifier_runs_setup(pv, inst, td)
        pv.set_srs(srs)
        for _ in range(2):
            p = pv.prove_dev(w.data_ptr(), 0)
        ref = ref or p
        assert p == ref, name
        dvp.lib.dvp_profile_reset(); dvp.lib.dvp_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            pv.prove_dev(w.data_ptr(), 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8 * 1e3
        dvp.lib.dvp_profile_enable(0)
        st = {}
        for k in ("msm_total", "msm_sort", "msm_affine_round0", "msm_affine_rest", "msm_tail"):
            ms, n = C.c_double(0), C.c_uint64(0)
            dvp.check(dvp.lib.dvp_profile_read(k.encode(), C.byref(ms), C.byref(n)))
            st[k] = ms.value / 8
        other = st["msm_total"] - sum(v for k, v in st.items() if k != "msm_total")
        gb = (pv.msm_table(0)[0] + pv.msm_table(1)[0]) / 1e9
        print(f"{name}: plans {pv.msm_plan(0)} {pv.msm_plan(1)} tables {gb:.1f} GB: proof {dt:.2f} ms | sort {st['msm_sort']:.2f} round0 {st['msm_affine_round0']:.2f} "
              f"later {st['msm_affine_rest']:.2f} merge+tail {st['msm_tail']:.2f} reducer {other:.2f}", flush=True)
        pv.close()
