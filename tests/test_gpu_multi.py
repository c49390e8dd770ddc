"""GPU tests that need MORE THAN ONE MI355X (-m gpu; skipped on a one-GPU box, which is what the builder's and the driver's
test boxes are -- the RCCL path is otherwise exercised only by the driver's N = 2/4/8 scaling runs of bench.py).

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


def _rank(rank, world, port, log_m, q):
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
    proof = dvp.distributed.prove_sharded(dvp.distributed.GpuBackend(pv, dev), w)
    q.put((rank, proof.to_bytes()))
    dist.barrier()
    dist.destroy_process_group()


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
