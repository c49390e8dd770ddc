"""CPU test of the N>1 path (gloo, world_size 2): the sharding / all-gather / combine orchestration of
dv-pari_amd.distributed.prove_sharded, with the GPU work replaced by an oracle-backed backend
(the orchestration is what is under test here; the GPU kernels are covered by the -m gpu tests)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    dist_mod = importlib.import_module("dv-pari_amd.distributed")
    for total in (0, 1, 7, 8, 1000, 6291456):
        for world in (1, 2, 3, 8):
            rs = [dist_mod.shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import pyref as o
    import c_oracle as co
    from util import to_limbs, pts_to_np

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist_mod = importlib.import_module("dv-pari_amd.distributed")

    trap = (0x1111 + (1 << 100), 0x2222 + (1 << 90), 0x3333 + (1 << 80))
    tree = o.FFTree(4)
    st = o.setup_srs_scalars(tree, o.TOY_ROWS, o.TOY_COEFFS, 2, trap)

    def pack(pt):
        t = torch.zeros(10, dtype=torch.int64)
        if pt is None:
            t[8] = 1
        else:
            t[:8] = torch.from_numpy(pts_to_np([pt])[0].view(np.int64))
        return t

    def unpack(t):
        if int(t[8]) & 0xFFFFFFFF:
            return None
        a = t[:8].numpy().view(np.uint64)
        return (int.from_bytes(a[:4].tobytes(), "little"), int.from_bytes(a[4:].tobytes(), "little"))

    class OracleBackend:
        """same protocol as distributed.GpuBackend, arithmetic by the C oracle"""

        def dims(self):
            return 8, 8  # n_wires, m of the toy instance

        def begin(self, assignment, need_extend=True):
            self.state = {}
            self.w = assignment
            self.pr = o.prove_scalars(tree, st, assignment[1:3], assignment[3:], self._alpha)
            gk = st["g_k"][0] + st["g_k"][1] + st["g_k"][2]
            self.sc = [self.pr["w"] + self.pr["q2"], self.pr["s_k"]]
            if not need_extend:  # a rank that skips the extends has no q2 and no k_r: poison them
                self.sc[0] = self.sc[0][:8] + [12345] * 8
                self.sc[1] = self.sc[1][:16] + [54321] * 16
            self.bases = [[co.k233_mulgen(k) for k in st["g_m"] + st["g_q"]], [co.k233_mulgen(k) for k in gk]]

        # the extends by vector (distributed.prove_sharded, from three extender ranks up): this rank's own extended vectors
        # are the oracle's, the others start as poison and must arrive by broadcast; the quotient recomputed from what
        # arrived must be the oracle's q2 / r2
        def extend_count(self):
            return 3

        def extend_vectors(self, vectors):
            self.vector_sharded = sorted(vectors)
            names = ("a2", "b2", "c2")
            self.ext = [torch.from_numpy(to_limbs(self.pr[names[v]] if v in vectors else [777] * 8).view(np.int64).copy()) for v in range(3)]

        def extended_tensor(self, v):
            return self.ext[v]

        def quotient(self):
            a2, b2, c2 = ([int.from_bytes(row.numpy().view(np.uint64).tobytes(), "little") for row in t] for t in self.ext)
            assert [a2, b2, c2] == [self.pr["a2"], self.pr["b2"], self.pr["c2"]], "an extended vector did not arrive"
            r2 = [(a2[i] * b2[i] - self.pr["i2"][i]) % o.P for i in range(8)]
            q2 = [(r2[i] - c2[i]) * st["tables"]["z_vals2inv"][i] % o.P for i in range(8)]
            assert r2 == self.pr["r2"] and q2 == self.pr["q2"]
            self.sc = [self.pr["w"] + q2, self.pr["s_k"]]  # un-poison: this rank now holds q2 and (through r2) k_r

        def _alpha(self, dl):
            return o.transcript_challenge(co.xsk233_encode(co.k233_mulgen(dl)), o.TOY_PUBLIC)

        def msm_size(self, which):
            return len(self.sc[which])

        def msm_partial(self, which, lo, hi):
            if hi == lo:
                return pack(None)
            b = [p for p in self.bases[which][lo:hi]]
            inf = np.array([1 if p is None else 0 for p in b], dtype=np.uint8)
            arr = pts_to_np([p if p is not None else (0, 0) for p in b])
            return pack(co.msm(to_limbs(self.sc[which][lo:hi]), arr, inf))

        def combine(self, gathered):
            acc = None
            for row in gathered:
                acc = o.k233_add(acc, unpack(row))
            return pack(acc)

        def challenge(self, point):
            self.commit = unpack(point)

        # the challenge phase sharded by index (distributed.prove_sharded uses it when world > 1): this rank's slice of the
        # three barycentric sums as a 128-byte record, then -- from the records of all ranks -- a0 b0 i0, which must equal the
        # oracle's, and only the K scalars of the rank's own MSM range (everything else poisoned)
        def challenge_partial(self, point, d_range, k_range):
            self.commit = unpack(point)
            tb = st["tables"]
            alpha = self.pr["alpha"]
            sums = []
            for y in (self.pr["a"], self.pr["b"], self.pr["i"]):
                acc = 0
                for i in range(d_range[0], d_range[1]):
                    acc = (acc + y[i] * tb["bar_wts"][i] % o.P * o.fr_inv((alpha - tb["D"][i]) % o.P)) % o.P
                sums.append(acc)
            rec = torch.zeros(16, dtype=torch.int64)
            rec[:12] = torch.from_numpy(to_limbs(sums).reshape(-1).view(np.int64))
            rec[12] = -1  # alpha-in-domain flag: none
            return rec

        def challenge_finish(self, gathered, k_range):
            assert gathered.shape == (world, 16) and all(int(row[12]) == -1 for row in gathered)
            tot = [0, 0, 0]
            for row in gathered:
                limbs = row[:12].numpy().view(np.uint64).reshape(3, 4)
                for v in range(3):
                    tot[v] = (tot[v] + sum(int(limbs[v][k]) << (64 * k) for k in range(4))) % o.P
            z = self.pr["z_alpha"]
            assert [t * z % o.P for t in tot] == [self.pr["a0"], self.pr["b0"], self.pr["i0"]]
            lo, hi = k_range
            self.sc[1] = [s if lo <= k < hi else 999 for k, s in enumerate(self.sc[1])]
            self.sharded_challenge = True

        def finish(self, point):
            return (co.xsk233_encode(self.commit), co.xsk233_encode(unpack(point)), self.pr["a0"], self.pr["b0"])

    be = OracleBackend()
    assignment = [1] + o.TOY_PUBLIC + o.TOY_PRIVATE
    proof = dist_mod.prove_sharded(be, assignment)
    exp_commit = co.xsk233_encode(co.k233_mulgen(be.pr["dl_commit_p"]))
    exp_kzg = co.xsk233_encode(co.k233_mulgen(be.pr["dl_kzg"]))
    ok = proof[0] == exp_commit and proof[1] == exp_kzg and getattr(be, "sharded_challenge", False)
    q.put((rank, ok, proof[0].hex(), getattr(be, "vector_sharded", None)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_prove_sharded_gloo(world):
    """world 2: uniform slices; world 3: one rank skips the extends (distributed.shard_plan) and owns only [w], [k_a|k_b];
    world 4: three extender ranks, so the extends are split by vector (one each, three broadcasts in the extender group);
    world 8 (the node size the bench is run at): three ranks skip the extends, five share them by vector, one-element
    shards, eight records per all-gather"""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert len({r[2] for r in res}) == 1  # every rank holds the same commitment
    # from three extender ranks up the extends are split by vector: every vector has exactly one owner among the extenders
    sys.path.insert(0, ROOT)
    dist_mod = importlib.import_module("dv-pari_amd.distributed")
    n_ext_ranks = sum(1 for pl in dist_mod.shard_plan(world, 8, 8) if pl[2])
    owned = sorted(v for r in res if r[3] is not None for v in r[3])
    assert owned == ([0, 1, 2] if n_ext_ranks >= 3 else []), (n_ext_ranks, res)
