"""GPU tests of the multi-process / multi-device paths (-m gpu).  On a one-GPU box -- what the builder's and the driver's
test boxes are -- RCCL runs in a one-rank group and `bench.py --gpus N` is rehearsed with every rank on cuda:0; the tests
that need MORE THAN ONE MI355X are skipped there (the exchange between devices is otherwise exercised only by the
driver's N = 2/4/8 scaling runs of bench.py).

  * two processes, one GPU each, torch.distributed over RCCL: prove_sharded must return the single-GPU proof bytes;
  * one process, dvp_set_devices([0, 1]): the in-library path over two real devices (peer copies over xGMI) -- same bytes.
"""
import importlib
import os
import random
import socket
import sys

import numpy as np
import pytest

import pyref as o

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus(dvp):
    return dvp.lib.dvp_device_count()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, log_m, q, always_gather=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    dvp = importlib.import_module("dv-pari_amd")
    dvp.check(dvp.lib.dvp_set_device(rank))
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    dev = torch.device("cuda", rank)
    w = torch.from_numpy(dvp.fr.vec([1] + pub + prv).view(np.int64)).to(dev)
    proof = dvp.distributed.prove_sharded(dvp.distributed.GpuBackend(pv, dev), w, always_gather=always_gather)
    q.put((rank, proof.to_bytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_collectives_execute_on_one_gpu(dvp):
    """RCCL for real on a one-GPU box: a one-rank nccl process group, prove_sharded(always_gather=True) -- both 80-byte
    all-gathers and the 128-byte record all-gather of the challenge phase go through RCCL kernels, the records through
    dvp_points_sum_dev / dvp_prove_challenge_finish -- and the proof bytes must equal the plain single-GPU proof.  (What a
    one-rank group cannot show is the exchange BETWEEN devices; that needs the two-GPU tests below.)"""
    import torch.multiprocessing as mp

    log_m = 13
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    ref = pv.prove(pub, prv).to_bytes()
    pv.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rank, args=(0, 1, _free_port(), log_m, q, True))
    p.start()
    got = dict([q.get(timeout=600)])
    p.join(timeout=120)
    assert p.exitcode == 0 and got[0] == ref


def test_bench_self_launch_rehearsal(dvp):
    """`python bench.py --gpus N` with nothing around it must start its own ranks, and rank 0's line must say what ran.
    Rehearsal knobs of a one-GPU box: every rank on cuda:0, gloo instead of RCCL; 4 ranks -- the most this box admits beside the
    test process itself and the in-library child that rank 0 starts at the end (at most 6 processes may use its GPU: a 5-rank run was
    killed by the guard; the 8-rank plan is walked on the CPU, tests/test_distributed_cpu.py) -- = a plan with one rank that skips
    the extends and three that split them by vector (three broadcasts of device buffers in the extender group).  The line carries the plan, the MEASURED costs it was made from (one timed extend, one timed broadcast, one
    timed MSM slice) and every rank's own stage times; the in-library leg (ms_per_step_inproc, dvp_set_devices over a repeated id)
    runs in the child."""
    import json
    import subprocess

    env = dict(os.environ, DVP_BENCH_SHARE_GPU="1", DVP_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    n = 4
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--log-m", "14"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["backend"] == "gloo" and len(d["ms_per_step_ranks"]) == n
    assert d["ms_per_step_inproc"] is not None and d["ms_per_step"] > 0, d.get("inproc_error")
    assert d["scaling_measured"] is True
    assert len(lines[0]) < 6000
    d = {**json.load(open(os.path.join(ROOT, d["detail"]))), **d}  # the line is the headline; plans and per-rank stages are in the sidecar it names
    plan = d["shard_plan"]
    assert [p["rank"] for p in plan] == list(range(n))
    m, n_wires = 1 << 14, d["config"]["n_wires"]
    for key, total in (("commit_msm_range", n_wires + m), ("k_msm_range", 4 * m)):  # the slices tile both MSMs
        cuts = sorted(tuple(p[key]) for p in plan if p[key][1] > p[key][0])
        assert cuts[0][0] == 0 and cuts[-1][1] == total and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    costs = d["shard_plan_costs"]
    assert costs["replicated"] > 0 and costs["split"] > 0 and costs["ms"]["extend_one_vector"] > 0 and costs["ms"]["broadcast_one_vector"] > 0
    assert len(d["stages_ms_per_step_by_rank"]) == n and all(r["msm_total_ms_per_step"] > 0 for r in d["stages_ms_per_step_by_rank"])
    # the replicas figure (N independent provers, measured after the timed loop) rides in the same line
    rep = d["replicas"]
    assert rep["n"] == n and rep["constraints_per_s"] > 0 and len(rep["ms_per_proof_each"]) == n


def test_bench_replicas_mode_rehearsal(dvp):
    """`python bench.py --gpus N --replicas`: N independent provers, one whole proof per rank and step, no data-path collective --
    the throughput mode DESIGN 7 recommends for N GPUs.  Rehearsed with two ranks on cuda:0 over gloo: the line says weak scaling,
    `value` is the aggregate of the ranks, and no sharded-mode keys (shard plan, in-library leg) are produced."""
    import json
    import subprocess

    env = dict(os.environ, DVP_BENCH_SHARE_GPU="1", DVP_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--replicas", "--steps", "3", "--warmup", "1", "--log-m", "12"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["rccl_ranks"] == 2 and "replicas" in d["config"]["sharding"]
    rep = d["replicas"]
    assert rep["n"] == 2 and len(rep["ms_per_proof_each"]) == 2
    # value = the ranks' proofs together over the slowest rank's time: between one and two ranks' own rates
    own = [(1 << 12) / (ms * 1e-3) for ms in rep["ms_per_proof_each"]]
    assert 0.5 * min(own) < d["value"] <= 2.05 * max(own), (d["value"], own)
    assert "ms_per_step_inproc" not in d and "replicas_error" not in d
    full = json.load(open(os.path.join(ROOT, d["detail"])))
    assert "shard_plan" not in full


@pytest.mark.parametrize("world", [2])
def test_two_ranks_over_rccl_same_bytes(dvp, world):
    if _n_gpus(dvp) < world:
        pytest.skip(f"needs {world} GPUs")
    import torch.multiprocessing as mp

    log_m = 14
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(log_m)
    td = dvp.srs.Trapdoor(0x1234567 + (1 << 200), 0x7654321 + (1 << 190), 0xABCDEF + (1 << 180))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    ref = pv.prove(pub, prv).to_bytes()
    pv.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, log_m, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(got[r] == ref for r in range(world))


def test_in_library_two_real_devices(dvp):
    if _n_gpus(dvp) < 2:
        pytest.skip("needs 2 GPUs")
    inst, pub, prv = dvp.gnark_r1cs.synthetic_dense(16)
    rnd = random.Random(216)
    td = dvp.srs.Trapdoor(rnd.randrange(1, o.P), rnd.randrange(1, o.P), rnd.randrange(1, o.P))
    dvp.check(dvp.lib.dvp_set_device(0))
    pv = dvp.proving.Prover(inst)
    pv.set_srs(dvp.srs.verifier_runs_setup(pv, inst, td))
    ref = pv.prove(pub, prv)
    try:
        dvp.set_devices(list(range(min(_n_gpus(dvp), 8))))
        assert pv.prove(pub, prv) == ref
    finally:
        dvp.set_devices([])
    assert dvp.srs.verify(td, pub, ref)
    pv.close()
