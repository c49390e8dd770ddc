#!/usr/bin/env python3
"""Per-phase time of ONE rank of a W-way sharded prove, measured on a single GPU (projection of bench.py --gpus W:
the ranks are symmetric, the exchange is one 80-byte all-gather per MSM and one of 128 bytes in the challenge phase).  python tools/shard_profile.py [log_m]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

dvp = importlib.import_module("dv-pari_amd")
log_m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
pv = dvp.proving.Prover(inst)
pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
assignment = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
be = dvp.distributed.GpuBackend(pv, dev)
# a second prover computes the FULL MSMs the challenge / finish phases need: a prover keeps its fixed-base tables for the
# index range it is asked for (a rank always asks for the same slice), so mixing slices and full ranges on one prover would
# rebuild them inside the timed calls
pv_full = dvp.proving.Prover(inst)
pv_full.set_srs(dvp.srs.verifier_runs_setup(pv_full, inst, td))
be_full = dvp.distributed.GpuBackend(pv_full, dev)
be_full.begin(assignment, True)
full_a = be_full.msm_partial(0, 0, be_full.msm_size(0)).clone()
be_full.challenge(full_a)
full_b = be_full.msm_partial(1, 0, be_full.msm_size(1)).clone()


def timed(f):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3


for world in [int(x) for x in os.environ.get("SHARD_WORLDS", "1,2,4,8").split(",")]:
    plan = dvp.distributed.shard_plan(world, *be.dims())
    roles = sorted({0, world - 1})  # rank 0 skips the extends when the plan has such ranks; the last rank never does
    worst = None
    for rank in roles:
        range_a, range_b, need = plan[rank]
        acc = None
        for rep in range(4):
            ph = {}
            ext_ranks = [r for r in range(world) if plan[r][2]]
            if need and len(ext_ranks) >= 3 and os.environ.get("SHARD_EXTENDS", "vector") != "replicated":
                # extends by vector (prove_sharded from three extender ranks up): this rank's own vectors + the quotient; the
                # broadcasts of the other vectors (2 x 32 MB in, <= 32 MB out per rank at 2^20) are NOT in a one-GPU projection
                own = [v for v in range(pv.extend_count()) if dvp.distributed.extend_owner(v, ext_ranks) == rank]
                def beg():
                    be.begin(assignment, False)
                    be.extend_vectors(own)
                    for v in range(pv.extend_count()):  # stand-in for the broadcasts (not in a one-GPU projection): whatever the buffers hold
                        if v not in own:
                            be.mark_extended(v)
                    be.quotient()
                _, ph["begin"] = timed(beg)
            else:
                _, ph["begin"] = timed(lambda: be.begin(assignment, need))
            _, ph["msmA_part"] = timed(lambda: be.msm_partial(0, *range_a))
            if world > 1 and os.environ.get("SHARD_CHALLENGE", "sharded") != "full":  # the index-sharded challenge: own slice of the sums, (stand-in for the all-gather: the own record
                # repeated -- the values do not matter for the timing), then the K scalars of the own range only
                def chal():
                    rec = be.challenge_partial(full_a, dvp.distributed.shard_range(be.dims()[1], rank, world), range_b)
                    be.challenge_finish(torch.stack([rec] * world), range_b)
                _, ph["challenge"] = timed(chal)
            else:
                _, ph["challenge"] = timed(lambda: be.challenge(full_a))
            _, ph["msmB_part"] = timed(lambda: be.msm_partial(1, *range_b))
            _, ph["finish"] = timed(lambda: be.finish(full_b))
            if rep:
                acc = ph if acc is None else {k: min(acc[k], v) for k, v in ph.items()}
        tot = sum(acc.values())
        tag = "extends" if need else "no extends"
        print(f"world {world} rank {rank} ({tag}): " + "  ".join(f"{k} {v:6.2f}" for k, v in acc.items()) + f"   sum {tot:6.2f} ms", flush=True)
        worst = tot if worst is None else max(worst, tot)
    print(f"world {world}: slowest rank {worst:6.2f} ms  ({(1 << log_m) / worst / 1e3:.1f} M constraints/s projected)", flush=True)
