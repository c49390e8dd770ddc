"""Multi-GPU prove: one process per GPU, torch.distributed over RCCL/xGMI.

Only the two MSMs shard (they are sums over disjoint index ranges): every rank runs the
cheap Fr stages redundantly on its own copy of the data, computes the partial sum of its contiguous
slice of (scalars, bases), and the partial points are combined with ONE all-gather of 80 bytes per
rank followed by a local N-term addition.  RCCL has no reduction operator for elliptic-curve points,
so "all-reduce of partial bucket sums" is realised as all-gather + local add; the message is
latency-bound (<= 640 B on 8 GPUs), xGMI bandwidth is irrelevant here.

The GPU work is injected through `backend` so the orchestration is testable on CPU with gloo
(tests/test_distributed_cpu.py uses an oracle-backed backend)."""
from dataclasses import dataclass


def shard_range(total: int, rank: int, world: int):
    """contiguous, balanced slices: the first (total % world) ranks get one extra element"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class GpuBackend:
    """Default backend: dv-pari_amd.proving.Prover phases + torch CUDA tensors."""
    prover: object
    device: object

    def __post_init__(self):
        import torch

        self.torch = torch
        self.part = torch.zeros(10, dtype=torch.int64, device=self.device)  # x||y (8 words) + inf flag (u32) + pad
        self.ones = None

    def begin(self, assignment):
        self.prover.begin(assignment.data_ptr(), self.torch.cuda.current_stream().cuda_stream)

    def msm_size(self, which):
        return self.prover.msm_size(which)

    def msm_partial(self, which, lo, hi):
        st = self.torch.cuda.current_stream().cuda_stream
        self.prover.msm_partial(which, lo, hi, self.part.data_ptr(), self.part.data_ptr() + 64, st)
        return self.part

    def combine(self, gathered):
        """gathered: [world, 10] int64 -> one point tensor [10] (sum of the partial points)"""
        from . import curve

        torch = self.torch
        world = gathered.shape[0]
        if self.ones is None or self.ones.shape[0] != world:
            self.ones = torch.zeros((world, 4), dtype=torch.int64, device=self.device)
            self.ones[:, 0] = 1
        xy = gathered[:, :8].contiguous()
        inf = (gathered[:, 8] & 0xFFFFFFFF).to(torch.uint8).contiguous()
        out = torch.zeros(10, dtype=torch.int64, device=self.device)
        curve.multi_scalar_mul_dev(self.ones.data_ptr(), xy.data_ptr(), inf.data_ptr(), world, out.data_ptr(),
                                   out.data_ptr() + 64, torch.cuda.current_stream().cuda_stream)
        return out

    def challenge(self, point):
        self.prover.challenge(point.data_ptr(), point.data_ptr() + 64, self.torch.cuda.current_stream().cuda_stream)

    def finish(self, point):
        return self.prover.finish(point.data_ptr(), point.data_ptr() + 64, self.torch.cuda.current_stream().cuda_stream)


def prove_sharded(backend, assignment, group=None):
    """Proof::prove (src/proving.rs:426-688) with both MSMs sharded over the ranks of `group`.
    Every rank returns the same proof."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    backend.begin(assignment)
    proof = None
    for which in (0, 1):
        lo, hi = shard_range(backend.msm_size(which), rank, world)
        part = backend.msm_partial(which, lo, hi)
        if world > 1:
            gathered = [torch.empty_like(part) for _ in range(world)]
            dist.all_gather(gathered, part, group=group)
            point = backend.combine(torch.stack(gathered))
        else:
            point = part
        if which == 0:
            backend.challenge(point)
        else:
            proof = backend.finish(point)
    return proof
