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
    """Drives one engine per rank.  Split form: `engine` needs mccfr_iteration(), mccfr_traverse(it, b0, nb), mccfr_apply();
    `all_reduce()` must sum the engine's bound delta buffer in place across ranks (no-op for world == 1).
    fused_exchange=True (a validated peer-memory exchange is connected, see connect_peer_exchange): the engine's
    mccfr_iterate_sharded(b0, nb, n) runs whole iterations in the library, the exchange inside the reduce+apply kernel."""

    def __init__(self, engine, rank, world, all_reduce=None, fused_exchange=False, always_exchange=False, replica_check=None,
                 check_every=4096):
        """replica_check: callable that raises when the ranks' tables differ (make_gpu_engine leaves one in ctx.replica_check).
        It is called every `check_every` iterations of run() when the connected exchange is the LIGHT protocol form, whose
        store ordering rests on measured behaviour rather than on fences (scopa_p2p.h): a rare stale row would otherwise
        corrupt the summed deltas silently.  The fenced form and the collective need no such check."""
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        self.all_reduce = all_reduce if all_reduce is not None else (lambda: None)
        self.fused_exchange, self.always_exchange = bool(fused_exchange), bool(always_exchange)
        self.replica_check = replica_check if replica_check is not None else getattr(engine, "replica_check", None)
        self.check_every = int(check_every)
        self._since_check = 0   # iterations through the light exchange form since the replicas were last compared -- carried ACROSS run() calls

    def iteration(self, batch_total):
        b0, nb = shard_range(batch_total, self.rank, self.world)
        if self.fused_exchange:
            self.engine.mccfr_iterate_sharded(b0, nb, 1)
            return
        it = self.engine.mccfr_iteration()
        self.engine.mccfr_traverse(it, b0, nb)
        if self.world > 1 or self.always_exchange:
            self.all_reduce()
        self.engine.mccfr_apply()

    def run(self, batch_total, n_iters):
        if self.fused_exchange:
            b0, nb = shard_range(batch_total, self.rank, self.world)
            guarded = self._guarded()
            left = int(n_iters)
            while left > 0:
                k = min(left, self.check_every - self._since_check) if guarded else left   # a driver that calls run() in short chunks is checked too
                self.engine.mccfr_iterate_sharded(b0, nb, k)
                left -= k
                if guarded:
                    self._since_check += k
                    if self._since_check >= self.check_every:
                        self.final_check()
            return
        for _ in range(int(n_iters)):
            self.iteration(batch_total)


    def _guarded(self):
        return self.replica_check is not None and getattr(self.engine, "exchange_form", None) == "light" and self.check_every > 0

    def final_check(self):
        """Compare the replicas' tables now if iterations have gone through the light exchange form since the last comparison (every rank must call it:
        the comparison is a collective).  run() calls it every `check_every` iterations; a caller that stops earlier -- bench.py at the end of its timed
        regions -- calls it itself."""
        if self._guarded() and self._since_check > 0:
            self._since_check = 0
            self.replica_check()


def connect_peer_exchange(ctx, rank, world, device, rounds=32, form="fenced"):
    """Set up the library's one-shot peer-memory all-reduce (scopa_p2p_*, include/scopa.h) between the `world` ranks of an
    initialised torch.distributed group (one process per GPU, one node) and PROVE it before use: `rounds` exchanges of
    rank-distinct random payloads must equal, bit for bit, the rank-ordered sum computed from an all-gather through
    torch.distributed (RCCL), on every rank, with no wait timing out.  form: "fenced" = the textbook protocol (system-scope
    release / acquire fences) -- correct by construction and the DEFAULT; "light" = sc0 sc1 accesses ordered by s_waitcnt alone
    (4.5 us less per iteration; rests on the fabric acknowledging a remote store only once it is visible, which the architecture
    does not promise: an explicit opt-in, and ShardedMCCFR.run then compares the replicas' tables periodically); "auto" = light
    if it passes the validation on this topology, else fenced (validated the same way).  Returns (ok, reason); every
    rank returns the same answer (each verdict is itself all-reduced), so the ranks switch paths together; the form in
    use is left in ctx.exchange_form."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import _lib
    if world > 1 and dist.get_backend() == "gloo":
        device = torch.device("cpu")      # handles and verdicts travel through the process group, whatever its backend

    def agree(ok):
        f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        if world > 1:
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item())

    def create_and_connect():
        reason, handle, ok = "", np.zeros(64, np.uint8), True
        try:
            handle = ctx.p2p_create(rank, world)
        except _lib.ScopaError as e:
            ok, reason = False, f"create: {e}"
        mine = torch.from_numpy(handle.copy()).to(device)
        parts = [torch.zeros(64, dtype=torch.uint8, device=device) for _ in range(world)]
        if world > 1:
            dist.all_gather(parts, mine)
        else:
            parts = [mine]
        if not agree(ok):
            return False, reason or "a peer could not create its inbox"
        try:
            ctx.p2p_connect(np.stack([p.cpu().numpy() for p in parts]))
        except _lib.ScopaError as e:
            ok, reason = False, f"connect: {e}"
        if not agree(ok):
            ctx.p2p_destroy()
            return False, reason or "a peer could not map the inboxes"
        return True, ""

    def validate():
        ok, reason = True, ""
        n = ctx.n_infosets * 5
        ctx.p2p_set_budget(2.0)
        for k in range(rounds):
            x = np.random.RandomState(1000 * k + rank).standard_normal(n) * 10.0 ** np.random.RandomState(k).randint(-6, 7)
            got = None
            try:                                   # a local failure must not skip the collectives below (the peers are in them)
                ctx.mccfr_delta_set(x.reshape(-1, 5))
                ctx.p2p_allreduce_delta()          # raises on a wait that gave up (SCOPA_ETIMEOUT)
                got = ctx.mccfr_delta_get().reshape(-1)
            except _lib.ScopaError as e:
                ok, reason = False, f"round {k}: {e}"
            xs = [torch.zeros(n, dtype=torch.float64, device=device) for _ in range(world)]
            if world > 1:
                dist.all_gather(xs, torch.from_numpy(x).to(device))
            else:
                xs = [torch.from_numpy(x).to(device)]
            want = xs[0].cpu().numpy().copy()
            for r in range(1, world):
                want = want + xs[r].cpu().numpy()          # rank order, one rounding per add: what p2p_exchange_wave4 does
            if got is not None and not np.array_equal(got, want):
                ok, reason = False, f"round {k}: sum differs from the rank-ordered reference"
        try:
            timeouts, _ = ctx.p2p_status()
            if timeouts:
                ok, reason = False, f"{timeouts} wait(s) timed out"
            ctx.mccfr_delta_set(np.zeros((ctx.n_infosets, 5)))
            ctx.p2p_set_budget(5.0)
        except _lib.ScopaError as e:
            ok, reason = False, f"status: {e}"
        return ok, reason

    why = ""
    for light in {"light": [True], "fenced": [False], "auto": [True, False]}[form]:
        ok, why = create_and_connect()
        if not ok:
            return False, why
        ctx.p2p_set_form(light)
        ok, why = validate()
        name = "light" if light else "fenced"
        if agree(ok):
            ctx.exchange_form = name
            return True, f"validated: {name} form, {rounds} exchanges bit-equal to the rank-ordered sum of an all-gather"
        why = f"{name} form: " + (why or "a peer failed validation")
        ctx.p2p_destroy()
    return False, why


import os as _os
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (hipIpcGetMemHandle / RCCL fail without it); must be set before the first HIP call


def make_gpu_engine(local_rank, perm16, seed, distributed=False, rank=0, exchange="rccl", exchange_form="fenced"):
    """Context on `local_rank` launching on a dedicated torch stream, with a torch-owned delta tensor bound as the
    all-reduce payload.  Returns (ctx, delta_tensor, stream, all_reduce).  distributed: a torch.distributed process group is
    up and the exchange step is wanted (its size is read from the group; a group of one rank is allowed, for timing the N > 1
    code path on one GPU).  exchange: "rccl" = torch.distributed.all_reduce; "p2p" = the library's one-shot peer-memory
    all-reduce (raises if it cannot be validated); "auto" = p2p if it validates on every rank, else rccl.  The path in use is
    recorded in `ctx.exchange`, the protocol form in `ctx.exchange_form`; `ctx.replica_check()` raises if the ranks' tables differ."""
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

    def replica_check():
        import hashlib
        import numpy as np
        R, S, _ = ctx.tables_get()
        h = np.frombuffer(hashlib.sha256(R.tobytes() + S.tobytes()).digest()[:8], dtype=np.uint8).copy()
        dev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(f"cuda:{local_rank}")
        mine = torch.from_numpy(h).to(dev)
        parts = [torch.zeros(8, dtype=torch.uint8, device=dev) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        if not all(bool((x == mine).all().item()) for x in parts):
            raise RuntimeError("ShardedMCCFR: the ranks' regret / strategy tables differ -- the delta exchange delivered different sums to different ranks")

    ctx.exchange, ctx.exchange_note, ctx.exchange_form = ("rccl" if distributed else "none"), "", None
    ctx.collective_all_reduce = all_reduce   # the torch.distributed path, kept reachable as the fallback when a connected peer exchange is dropped later
    ctx.replica_check = replica_check if distributed else None
    if distributed and exchange in ("auto", "p2p"):
        ok, why = connect_peer_exchange(ctx, rank, dist.get_world_size(), torch.device(f"cuda:{local_rank}"), form=exchange_form)
        ctx.exchange_note = why
        if ok:
            ctx.exchange = "p2p"
            return ctx, delta, stream, ctx.p2p_allreduce_delta
        if exchange == "p2p":
            raise RuntimeError(f"peer-memory exchange unavailable: {why}")
        import sys
        print(f"[scopa_amd rank {rank} / cuda:{local_rank}] peer-memory exchange not used, falling back to torch.distributed all-reduce: {why}", file=sys.stderr, flush=True)
    return ctx, delta, stream, (all_reduce if distributed else (lambda: None))


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
