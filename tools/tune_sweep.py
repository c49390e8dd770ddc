"""A/B sweep of tuning knobs on the 2^20 prove: python tools/tune_sweep.py KNOB=v1,v2,... [KNOB2=...]  (cartesian).
Knobs that are read when a fixed-base context is built (window size, table flavour) get one prover per value; the
run-time knobs are swept on that prover in interleaved rounds (REPS rounds of 6 proofs per combination, median
reported) so that clock / thermal drift of the box does not masquerade as a knob effect.  Every proof is byte-compared."""
import importlib, itertools, os, statistics, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
log_m = int(os.environ.get("LOG_M", "20"))
reps = int(os.environ.get("REPS", "5"))
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
CREATE = {"DVP_MSM_FIXED_C", "DVP_MSM_ALIGNED_SIGNED", "DVP_FX_HI", "DVP_MSM_FIXED_MIN"}
knobs = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in sys.argv[1:]]
outer = [(k, v) for k, v in knobs if k in CREATE]
inner = [(k, v) for k, v in knobs if k not in CREATE]
w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).cuda()
ref = None
for oc in itertools.product(*[v for _, v in outer]):
    okw = {k: v for (k, _), v in zip(outer, oc)}
    with dvp.tune(**okw):
        pv = dvp.proving.Prover(inst)
        pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
        pv.prove_dev(w.data_ptr(), 0)  # builds the fixed-base contexts under the creation-time knobs
    combos = [{k: v for (k, _), v in zip(inner, ic)} for ic in itertools.product(*[v for _, v in inner])]
    times = [[] for _ in combos]
    for r in range(reps + 1):  # round 0 warms up
        for ci, kw in enumerate(combos):
            with dvp.tune(**okw, **kw):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(6):
                    p = pv.prove_dev(w.data_ptr(), 0)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
            if ref is None: ref = p
            assert p == ref
            if r: times[ci].append(dt * 1e3)
    for kw, ts in zip(combos, times):
        print({**okw, **kw}, "median %.2f ms  (min %.2f max %.2f)" % (statistics.median(ts), min(ts), max(ts)), flush=True)
    pv.close()
