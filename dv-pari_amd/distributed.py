"""Multi-GPU prove: one process per GPU, torch.distributed over RCCL/xGMI.

Only the two MSMs shard (they are sums over disjoint index ranges): every rank runs the
cheap Fr stages redundantly on its own copy of the data, computes the partial sum of its contiguous
slice of (scalars, bases), and the partial points are combined with ONE all-gather of 80 bytes per
rank followed by a local N-term addition.  The slices are not uniform (shard_plan): the scalars [w] of the first
MSM and [k_a | k_b] of the second need no extended evaluations, so the ranks that own only those skip the three
ECFFT extends and take a larger share of the MSM work in exchange.  RCCL has no reduction operator for elliptic-curve points,
so "all-reduce of partial bucket sums" is realised as all-gather + local add; the message is
latency-bound (<= 640 B on 8 GPUs), xGMI bandwidth is irrelevant here.

The challenge phase between the two MSMs is sharded by index as well: each rank inverts 1/(d - alpha) only where it
needs it, sums its slice of the three barycentric sums, the ranks all-gather 128-byte records (partial sums + the
alpha-in-domain flag), and each rank then forms only the K scalars of its own MSM range.

The GPU work is injected through `backend` so the orchestration is testable on CPU with gloo
(tests/test_distributed_cpu.py uses an oracle-backed backend)."""
from dataclasses import dataclass


def shard_range(total: int, rank: int, world: int):
    """contiguous, balanced slices: the first (total % world) ranks get one extra element"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_plan(world: int, n_wires: int, m: int, extend_pairs=None):
    """Per-rank (range of MSM 0, range of MSM 1, need_extend).  MSM 0 runs over [w (n_wires) | q2 (m)], MSM 1 over
    [k_a (m) | k_b (m) | k_r (2m)]; q2 and k_r need the extends (q2 itself, r2 inside k_r), the rest does not.  With s
    "extender" ranks (the last s) and cost E of the extends expressed in (scalar, base) pairs, the non-extenders take X
    pairs of the extend-free part and the extenders everything else; X equalises the two loads where the caps allow, and
    s minimises the larger load.  s == world (uniform slices, every rank extends) is what small worlds get.
    extend_pairs: None = the typed-in defaults below; a number = that cost for every s; a dict {"replicated": pairs,
    "split": pairs} = what measure_plan_costs MEASURED on this machine (every rank must pass the same values)."""
    ta, tb = n_wires + m, 4 * m
    cap_a, cap_b = n_wires, 2 * m
    # defaults, measured on ONE MI355X (tools/shard_profile.py): R1CS evaluation + three extends + quotient of 2^20 = 1.58 ms, the
    # time the sharded MSMs take for ~0.42 * 2^20 (scalar, base) pairs.  From three extenders up the extends are split by vector
    # (prove_sharded): an extender then spends 0.25 ms (quotient only) to 0.9 ms (one vector + quotient) plus the broadcasts
    # of the vectors it does not own (64-96 MB in over xGMI, which a one-GPU box cannot time): charged as 0.30 * m pairs.
    # bench.py --gpus N replaces both by measure_plan_costs (one timed extend, one timed broadcast, one timed MSM slice).
    best = None
    for s in range(world, 0, -1):
        if extend_pairs is None:
            e = (0.42 if s < 3 else 0.30) * m
        elif isinstance(extend_pairs, dict):
            e = float(extend_pairs["replicated" if s < 3 else "split"])
        else:
            e = float(extend_pairs)
        if s == world:
            x, load = 0.0, e + (ta + tb) / world
        else:
            x = min(float(cap_a + cap_b), (world - s) * (ta + tb + s * e) / world)
            load = max(x / (world - s), e + (ta + tb - x) / s)
        if best is None or load < best[0] * 0.98:  # prefer more extenders (simpler) unless the gain is real
            best = (load, s, x)
    _, s, x = best
    plan = []
    if s == world:
        for r in range(world):
            plan.append((shard_range(ta, r, world), shard_range(tb, r, world), True))
        return plan
    x_a = min(cap_a, int(round(x * cap_a / (cap_a + cap_b))))
    x_b = min(cap_b, int(round(x)) - x_a)
    n0 = world - s
    for r in range(world):
        if r < n0:
            plan.append((shard_range(x_a, r, n0), shard_range(x_b, r, n0), False))
        else:
            lo_a, hi_a = shard_range(ta - x_a, r - n0, s)
            lo_b, hi_b = shard_range(tb - x_b, r - n0, s)
            plan.append(((x_a + lo_a, x_a + hi_a), (x_b + lo_b, x_b + hi_b), True))
    return plan


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

    def begin(self, assignment, need_extend=True):
        self.prover.begin(assignment.data_ptr(), self.torch.cuda.current_stream().cuda_stream, need_extend)

    # ---- the extends by vector (SURVEY 8e option A): extend only `vectors`, hand out views of the extended vectors for the
    # exchange, then the quotient ------------------------------------------------------------------------------------------
    def extend_count(self):
        return self.prover.extend_count()

    def extend_vectors(self, vectors):
        mask = 0
        for v in vectors:
            mask |= 1 << v
        self.prover.extend_vectors(mask, self.torch.cuda.current_stream().cuda_stream)

    def extended_tensor(self, v):
        """zero-copy int64 [m, 4] view of extended vector v inside the prover's buffers (what the broadcast sends / fills)"""
        if getattr(self, "_ext_views", None) is None:
            self._ext_views = {}
        if v not in self._ext_views:
            class _View:  # __cuda_array_interface__: torch wraps the device address without copying
                pass
            view = _View()
            view.__cuda_array_interface__ = {"shape": (self.prover.m, 4), "typestr": "<i8", "data": (self.prover.extended_ptr(v), False), "version": 2}
            self._ext_views[v] = self.torch.as_tensor(view, device=self.device)
        return self._ext_views[v]

    def mark_extended(self, v):
        """extended vector v has been received into extended_tensor(v) (dvp_prove_mark_extended): the quotient refuses to run
        until every vector is extended here or marked"""
        self.prover.mark_extended(v)

    def quotient(self):
        self.prover.quotient(self.torch.cuda.current_stream().cuda_stream)

    def dims(self):
        """(n_wires, m) of the instance: what shard_plan needs"""
        return self.prover.inst.n_wires, self.prover.m

    def msm_size(self, which):
        return self.prover.msm_size(which)

    def msm_partial(self, which, lo, hi):
        st = self.torch.cuda.current_stream().cuda_stream
        self.prover.msm_partial(which, lo, hi, self.part.data_ptr(), self.part.data_ptr() + 64, st)
        return self.part

    def msm_partial_one_shot(self, which, lo, hi):
        """the same slice through the one-shot path (no fixed-base table is built or replaced): measure_plan_costs' probe"""
        from ._native import tune

        with tune(DVP_MSM_FIXED_MIN=1 << 40):
            return self.msm_partial(which, lo, hi)

    def combine(self, gathered):
        """gathered: [world, 10] int64 -- the all-gathered 80-byte records (x || y, u32 infinity flag, pad) -- -> one point
        tensor [10]: their sum by world - 1 additions (dvp_points_sum_dev)"""
        from ._native import lib, check

        torch = self.torch
        g = gathered.contiguous()
        out = torch.zeros(10, dtype=torch.int64, device=self.device)
        check(lib.dvp_points_sum_dev(g.data_ptr(), g.shape[0], out.data_ptr(), out.data_ptr() + 64, torch.cuda.current_stream().cuda_stream),
              "dvp_points_sum_dev")
        return out

    def challenge(self, point):
        self.prover.challenge(point.data_ptr(), point.data_ptr() + 64, self.torch.cuda.current_stream().cuda_stream)

    def challenge_partial(self, point, d_range, k_range):
        """this rank's share of the challenge phase -> its 128-byte record (16 int64 words) for the all-gather"""
        if getattr(self, "rec", None) is None:
            self.rec = self.torch.zeros(16, dtype=self.torch.int64, device=self.device)
        self.prover.challenge_partial(point.data_ptr(), point.data_ptr() + 64, d_range, k_range, self.rec.data_ptr(),
                                      self.torch.cuda.current_stream().cuda_stream)
        return self.rec

    def challenge_finish(self, gathered, k_range):
        g = gathered.contiguous()
        self.prover.challenge_finish(g.data_ptr(), g.shape[0], k_range, self.torch.cuda.current_stream().cuda_stream)

    def finish(self, point):
        return self.prover.finish(point.data_ptr(), point.data_ptr() + 64, self.torch.cuda.current_stream().cuda_stream)


import weakref

_WORLD = type("_DefaultGroup", (), {})()  # stands for group=None (the default process group) as a cache key


class _GroupCache:
    """per-process-group cache keyed by the group OBJECT (weakly): an entry dies with its group, so a new group that happens
    to reuse the id() of a destroyed one can never inherit its collective mode or extender sub-group"""

    def __init__(self):
        self._d = weakref.WeakKeyDictionary()

    def slot(self, group):
        key = _WORLD if group is None else group
        try:
            return self._d.setdefault(key, {})
        except TypeError:  # a group type that cannot be weakly referenced: no caching, every call decides afresh
            return {}

    def clear(self):
        self._d = weakref.WeakKeyDictionary()


_GATHER_MODE = _GroupCache()  # process group -> {"mode": "tensor" | "list"}, decided ONCE by probe_collectives


def probe_collectives(device, group=None):
    """Decide once per process group, in a throw-away exchange every rank makes at the same point (bench.py calls it before
    the warm-up; prove_sharded calls it on first use), whether the backend has all_gather_into_tensor (RCCL does) or only
    the list form.  Only "this collective does not exist here" is accepted as a reason to fall back, and every rank sees
    the same answer because the capability is a property of the backend, not of a rank; from then on a RuntimeError of a
    collective (a communicator fault) propagates, so all ranks fail loudly instead of one of them silently switching to a
    different collective than its peers."""
    import torch
    import torch.distributed as dist

    slot = _GATHER_MODE.slot(group)
    if "mode" in slot:
        return slot["mode"]
    world = dist.get_world_size(group)
    probe = torch.zeros(10, dtype=torch.int64, device=device)
    out = torch.empty(world * 10, dtype=torch.int64, device=device)  # flat: the one output shape every backend accepts
    mode = "tensor"
    try:
        dist.all_gather_into_tensor(out, probe, group=group)
    except NotImplementedError:
        mode = "list"
    except RuntimeError as e:
        msg = str(e).lower()
        if "not implemented" in msg or "not supported" in msg or "no backend" in msg or "unsupported" in msg:
            mode = "list"
        else:
            raise
    if mode == "list":
        dist.all_gather([torch.empty_like(probe) for _ in range(world)], probe, group=group)
    slot["mode"] = mode
    return mode


def measure_plan_costs(backend, assignment, group=None, reps: int = 3):
    """What shard_plan charges an extender rank, MEASURED on this machine instead of typed in: one timed begin with and without
    the extends, one timed single-vector extend + quotient, one timed broadcast of an m-vector inside `group` (the exchange of
    the vector split), and one timed slice of the K MSM (the exchange rate between milliseconds and (scalar, base) pairs).
    Collective: every rank of `group` calls it at the same point (bench.py: before the warm-up); the ranks agree on the MAXIMUM
    of each figure.  Returns {"replicated": pairs, "split": pairs, "ms": {...}} -- pass it to prove_sharded(plan_costs=..)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_wires, m = backend.dims()
    dev = assignment.device

    def timed(fn):
        fn()
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None or t < best else best
        return best

    n_ext = backend.extend_count()
    t_full = timed(lambda: backend.begin(assignment, True))
    t_none = timed(lambda: backend.begin(assignment, False))

    # one vector's extend ALONE between the events (begin is enqueued once, before them): no difference of two noisy timings,
    # which at small sizes came out <= 0 and skewed the plan
    FLOOR_MS = 1e-3
    backend.begin(assignment, False)
    t_one = max(FLOOR_MS, timed(lambda: backend.extend_vectors([0])))
    # quotient alone = full - none - n_ext vectors (the batched extend shares its constants, so this is a slight over-estimate)
    t_quot = max(FLOOR_MS, t_full - t_none - n_ext * t_one)
    t_bcast = 0.0
    if world > 1:
        buf = backend.extended_tensor(0)

        def bc():
            dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        t_bcast = timed(bc)
    lo, hi = shard_range(4 * m, rank, world)
    backend.begin(assignment, True)
    zero_pt = torch.zeros(10, dtype=torch.int64, device=dev)
    zero_pt[8] = 1  # the point at infinity as a commitment: alpha is then the transcript of the zero encoding -- any alpha serves a timing
    backend.challenge(zero_pt) if hasattr(backend, "challenge") else None
    # the slice goes through the ONE-SHOT path: the fixed-base path would build a table for this uniform slice that prove_sharded
    # throws away as soon as the plan gives the rank another range (with --warmup 0 that rebuild landed in the timed loop); the
    # plan only needs the exchange rate between milliseconds and pairs, and `one_shot_over_fixed` (measured at 2^20: 25.5 / 19.0 ms
    # per proof, MSM share) brings it to the fixed-base path's
    one_shot_over_fixed = 1.4
    if hasattr(backend, "msm_partial_one_shot"):
        t_msm = timed(lambda: backend.msm_partial_one_shot(1, lo, hi)) / one_shot_over_fixed
    else:
        t_msm = timed(lambda: backend.msm_partial(1, lo, hi))
    pairs_per_ms = (hi - lo) / t_msm if t_msm > 0 else float(m)
    fig = torch.tensor([t_full, t_none, t_one, t_quot, t_bcast, 1.0 / pairs_per_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(fig, op=dist.ReduceOp.MAX, group=group)
    t_full, t_none, t_one, t_quot, t_bcast, ms_per_pair = (float(x) for x in fig.tolist())
    ext_ranks_guess = max(3, min(world, n_ext))
    own = -(-n_ext // ext_ranks_guess)  # vectors an extender extends itself under the split
    split_ms = max(FLOOR_MS, own * t_one + (n_ext - own) * t_bcast + t_quot)
    return {"replicated": (t_full - t_none) / ms_per_pair, "split": split_ms / ms_per_pair,
            "ms": {"begin_with_extends": t_full, "begin_without": t_none, "extend_one_vector": t_one, "quotient": t_quot,
                   "broadcast_one_vector": t_bcast, "msm_ms_per_million_pairs": ms_per_pair * 1e6, "split_extender_extra": split_ms},
            "note": "measured once per group (max over ranks); shard_plan charges an extender rank `replicated` pairs when fewer than three ranks "
                    "extend and `split` pairs under the vector split"}


def _all_gather_records(part, world, group):
    """[world, k] tensor of every rank's record: ONE collective into one preallocated tensor where the backend has
    all_gather_into_tensor (RCCL does), the list form + stack otherwise (decided by probe_collectives, never mid-run)"""
    import torch
    import torch.distributed as dist

    if probe_collectives(part.device, group) == "tensor":
        out = torch.empty(world * part.numel(), dtype=part.dtype, device=part.device)
        dist.all_gather_into_tensor(out, part.reshape(-1), group=group)
        return out.view((world,) + tuple(part.shape))
    gathered = [torch.empty_like(part) for _ in range(world)]
    dist.all_gather(gathered, part, group=group)
    return torch.stack(gathered)


_EXT_GROUPS = _GroupCache()  # parent group -> {extender ranks: process group of the extender ranks}


def _extender_group(ext_ranks, group):
    """the sub-group the extended vectors are exchanged in; dist.new_group is collective over the PARENT group, so every
    rank calls this at the same point of its first sharded proof (the result is cached)"""
    import torch.distributed as dist

    slot = _EXT_GROUPS.slot(group)
    key = tuple(ext_ranks)
    if key not in slot:
        # new_group takes GLOBAL ranks; ext_ranks are ranks of `group`
        glob = [dist.get_global_rank(group, r) for r in ext_ranks] if group is not None else list(ext_ranks)
        slot[key] = dist.new_group(ranks=glob) if len(ext_ranks) < dist.get_world_size(group) else group
    return slot[key]


def forget_groups():
    """drop every cached collective mode / extender sub-group (call after dist.destroy_process_group(): the default group is
    keyed by a module-level stand-in that outlives it)"""
    _GATHER_MODE.clear()
    _EXT_GROUPS.clear()


def extend_owner(v, ext_ranks):
    """the extender rank that computes extended vector v (a, b, c' [, i] round-robin over the extender ranks)"""
    return ext_ranks[v % len(ext_ranks)]


def prove_sharded(backend, assignment, group=None, always_gather=False, shard_extends=True, plan_costs=None):
    """Proof::prove (src/proving.rs:426-688) with both MSMs sharded over the ranks of `group`.
    Every rank returns the same proof.  always_gather runs the collectives and the record combination even in a
    one-rank group (how a one-GPU box executes the RCCL calls of this path for real).
    shard_extends: the ranks that need q2 / r2 (shard_plan's extenders) split the three extends by VECTOR -- each extends
    the vectors it owns, one broadcast per vector inside the extender group hands them round, then every extender forms
    the quotient (SURVEY 8e option A; DESIGN.md section 7 has the byte / latency budget)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_wires, m = backend.dims()
    plan = shard_plan(world, n_wires, m, extend_pairs=plan_costs)  # plan_costs: measure_plan_costs' figures (the same on every rank) or None
    range_a, range_b, need_extend = plan[rank]
    ext_ranks = [r for r in range(world) if plan[r][2]]
    # worth it from three extenders up (one vector each): with two, the rank that owns two vectors saves one extend
    # (~0.5 ms at 2^20) and pays for it in two 32 MB sends over the one xGMI link it shares with its peer
    by_vector = shard_extends and len(ext_ranks) >= 3 and hasattr(backend, "extend_vectors")
    if by_vector:
        ext_group = _extender_group(ext_ranks, group)  # collective over `group` the first time: before any rank branches
    if by_vector and need_extend:
        backend.begin(assignment, False)
        n_ext = backend.extend_count()
        backend.extend_vectors([v for v in range(n_ext) if extend_owner(v, ext_ranks) == rank])
        work = [dist.broadcast(backend.extended_tensor(v), src=dist.get_global_rank(group, extend_owner(v, ext_ranks)) if group is not None
                               else extend_owner(v, ext_ranks), group=ext_group, async_op=True) for v in range(n_ext)]
        for v, wk in enumerate(work):
            wk.wait()
            if extend_owner(v, ext_ranks) != rank and hasattr(backend, "mark_extended"):
                backend.mark_extended(v)
        backend.quotient()
    else:
        backend.begin(assignment, need_extend)
    assert backend.msm_size(0) == n_wires + m and backend.msm_size(1) == 4 * m
    proof = None
    for which in (0, 1):
        lo, hi = range_a if which == 0 else range_b
        part = backend.msm_partial(which, lo, hi)
        if world > 1 or (always_gather and dist.is_initialized()):
            point = backend.combine(_all_gather_records(part, world, group))
        else:
            point = part
        if which == 0:
            if (world > 1 or (always_gather and dist.is_initialized())) and hasattr(backend, "challenge_partial"):
                # pointwise stages, batch inversions and barycentric sums by index: every rank sums its slice
                # of D, one all-gather of 128-byte records, then only the K scalars of its own MSM range
                rec = backend.challenge_partial(point, shard_range(m, rank, world), range_b)
                backend.challenge_finish(_all_gather_records(rec, world, group), range_b)
            else:
                backend.challenge(point)
        else:
            proof = backend.finish(point)
    return proof
