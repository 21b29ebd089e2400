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


# ---- SDCFR data parallelism (BASELINE configs[4]) ---------------------------------------------------------------------
def allreduce_gradients(parameters, world, all_reduce):
    """Average the gradients of `parameters` over `world` ranks with ONE flat all-reduce (13 776 float32 = 55 104 B for the
    34-128-64-16 advantage MLP): every rank then takes the identical Adam step, so the replicas' nets stay identical.
    `all_reduce(tensor)` must sum in place across ranks (torch.distributed.all_reduce on RCCL / gloo)."""
    import torch
    params = [p for p in parameters if p.grad is not None]
    if world <= 1 or not params:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    all_reduce(flat)
    flat /= float(world)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def broadcast_parameters(module, broadcast):
    """Make rank 0's initial weights everyone's (`broadcast(tensor)` = in-place broadcast from rank 0)."""
    for t in list(module.parameters()) + list(module.buffers()):
        broadcast(t.data)
