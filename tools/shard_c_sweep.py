"""fixed-base MSM of n = 2^lg points (the shard sizes of a W-way split) under every window size, aligned signed and aligned tau-adic
tables: python tools/shard_c_sweep.py [lg ...]  -- checks the cost model's own choice (first line of each block) against the sweep"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
dvp = importlib.import_module("dv-pari_amd")
rng = np.random.default_rng(7)
def rand_fr(n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); a[:, 3] &= (1 << 38) - 1
    return a
for lg in [int(x) for x in sys.argv[1:]] or [18, 19, 20]:
    n = 1 << lg
    bases, _ = dvp.curve.point_scalar_mul_gen_batch(rand_fr(n))
    s = rand_fr(n)
    for name, knobs in (("signed aligned", {}), ("tau-adic aligned", {"DVP_MSM_ALIGNED_SIGNED": 0})):
        ref = None
        for c in [0] + list(range(14, 22)):
            kn = dict(knobs); 
            if c: kn["DVP_MSM_FIXED_C"] = c
            with dvp.tune(**kn):
                fb = dvp.curve.FixedBaseMsm(bases)
                plan = fb.plan()
                if c and plan[0] != c:
                    fb.close(); continue
                d_s = torch.from_numpy(s.view(np.int64)).cuda()
                out = torch.zeros(10, dtype=torch.int64, device="cuda")
                run = lambda: dvp.check(dvp.lib.dvp_msm_ctx_run_dev(fb._h, d_s.data_ptr(), 0, n, out.data_ptr(), out.data_ptr() + 64, 0), "run")
                for _ in range(3): run()
                torch.cuda.synchronize()
                if ref is None: ref = out.clone()
                assert torch.equal(ref, out)
                t0 = time.perf_counter()
                for _ in range(10): run()
                torch.cuda.synchronize()
                print(f"2^{lg} {name} {'model' if not c else 'c=%d' % c}: plan {plan} {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
                fb.close()
