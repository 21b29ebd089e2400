"""Data-parallel batched MCCFR: one process per GPU, ONE sum-all-reduce per iteration.

The reference has no distributed code; this is the build's own N>1 path.  Traversals of
an iteration are independent given the iteration's frozen tables, and every random draw is keyed by the GLOBAL
traversal id, so the ids [0, batch_total) are simply partitioned over ranks.  Each rank accumulates its
[n_infosets][5] float64 delta (4 regret deltas + traverser-visit count), the deltas are summed over ranks
(RCCL over xGMI via torch.distributed backend "nccl"; gloo on CPU in tests), and every rank applies the identical
sum, so the replicas' tables stay identical without a broadcast.
"""


def shard_range(batch_total, rank, world):
    """Contiguous slice [b0, b0+nb) of the global traversal ids owned by `rank`; remainders go to the low ranks."""
    base, rem = divmod(int(batch_total), int(world))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


class ShardedMCCFR:
    """Drives one engine per rank.  `engine` needs mccfr_iteration(), mccfr_traverse(it, b0, nb), mccfr_apply();
    `all_reduce()` must sum the engine's bound delta buffer in place across ranks (no-op for world == 1)."""

    def __init__(self, engine, rank, world, all_reduce=None):
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        self.all_reduce = all_reduce if all_reduce is not None else (lambda: None)

    def iteration(self, batch_total):
        it = self.engine.mccfr_iteration()
        b0, nb = shard_range(batch_total, self.rank, self.world)
        self.engine.mccfr_traverse(it, b0, nb)
        if self.world > 1:
            self.all_reduce()
        self.engine.mccfr_apply()

    def run(self, batch_total, n_iters):
        for _ in range(int(n_iters)):
            self.iteration(batch_total)


def make_gpu_engine(local_rank, perm16, seed, world=1):
    """Context on `local_rank` launching on a dedicated torch stream, with a torch-owned delta tensor bound as the
    all-reduce payload.  Returns (ctx, delta_tensor, stream, all_reduce)."""
    import torch
    import torch.distributed as dist
    from . import _lib
    torch.cuda.set_device(local_rank)
    stream = torch.cuda.Stream(device=local_rank)
    ctx = _lib.Context(local_rank, stream=stream.cuda_stream)
    n_inf = ctx.set_deal(perm16)
    ctx.mccfr_seed(seed)
    with torch.cuda.stream(stream):
        delta = torch.zeros((n_inf, 5), dtype=torch.float64, device=f"cuda:{local_rank}")
    stream.synchronize()
    ctx.mccfr_bind_delta(delta.data_ptr(), delta.numel() * 8)

    def all_reduce():
        with torch.cuda.stream(stream):
            dist.all_reduce(delta, op=dist.ReduceOp.SUM)

    return ctx, delta, stream, (all_reduce if world > 1 else (lambda: None))
