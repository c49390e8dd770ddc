"""How the first pair round's 64-byte gathers are served: table in ordinary (cached) or uncached device memory x ordinary or
non-temporal loads.  rocprofv3's TCC counters show every read request of that round as a 128-byte one -- half of each
fetched line is never used -- so the request granularity is worth an experiment.  Per mode: the 2^20 proof time, the
first-round launch time (HIP events) and dvp_ubench_gather on the K-MSM's table.  python tools/gather_modes.py"""
import ctypes as C, importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_m = int(os.environ.get("LOG_M", "20"))
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
srs = None
ref = None
for slide in (1, 0):
    for unc in (0, 1):
        for nt in (0, 1):
            with dvp.tune(DVP_MSM_SLIDE=slide, DVP_MSM_TABLE_UNCACHED=unc, DVP_MSM_NT_LOADS=nt):
                pv = dvp.proving.Prover(inst)
                if srs is None:
                    srs = dvp.srs.verifier_runs_setup(pv, inst, td)
                pv.set_srs(srs)
                for _ in range(2):
                    p = pv.prove_dev(w.data_ptr(), 0)
                if ref is None:
                    ref = p
                assert p == ref
                dvp.lib.dvp_profile_reset(); dvp.lib.dvp_profile_enable(1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(8):
                    pv.prove_dev(w.data_ptr(), 0)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8 * 1e3
                dvp.lib.dvp_profile_enable(0)
                ms, n = C.c_double(0), C.c_uint64(0)
                dvp.check(dvp.lib.dvp_profile_read(b"msm_affine_round0", C.byref(ms), C.byref(n)))
                tp, tb, r = C.c_void_p(0), C.c_uint64(0), C.c_double(0)
                dvp.check(dvp.lib.dvp_prover_msm_table_ptr(pv._h, 1, C.byref(tp), C.byref(tb)))
                dvp.check(dvp.lib.dvp_ubench_gather(tp, tb.value, 4, C.byref(r)))
                print(f"slide {slide} uncached {unc} nt {nt}: proof {dt:.2f} ms, first rounds {ms.value / 8:.2f} ms per proof, table {tb.value / 1e9:.1f} GB, "
                      f"ubench {r.value / 1e9:.1f} G gathers/s", flush=True)
                pv.close()
