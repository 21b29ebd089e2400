"""The N>1 path on CPU: world_size-2 gloo processes drive scopa_amd.distributed.ShardedMCCFR.

There is no GPU here, so each rank's engine is a test double whose traverse step is the ORACLE (allowed: tests may
call the oracle as the checker).  What is under test is the product's host logic: the partition of global traversal
ids, the single sum-all-reduce of the [n_infosets][5] delta per iteration, and every rank applying the same sum --
i.e. that 2 ranks reproduce the 1-rank result (integer visit counts exactly, float64 sums to 1e-12)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


class OracleEngine:
    """Same three calls as scopa_amd._lib.Context, computed by the oracle; delta layout [I][5] like the device buffer."""

    def __init__(self, seed):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        self.t = O.Tree(seed=42)
        self.R, self.S, _ = self.t.tables()
        self.delta = np.zeros((self.t.n_infosets, 5))
        self.seed, self.it = seed, 0

    def mccfr_iteration(self):
        return self.it

    def mccfr_traverse(self, iteration, b0, nb):
        dR, dS, _, _ = self.t.mccfr_batched_delta(self.R, self.seed, iteration, b0, nb)
        self.delta[:, :4] += dR
        self.delta[:, 4] += np.rint(dS.sum(1))

    def mccfr_apply(self):
        # k_mccfr_apply: regret += delta; strategy += count * sigma(frozen regret); delta <- 0
        pos = np.maximum(self.R, 0)
        s = pos.sum(1, keepdims=True)
        n = self.t.infoset_nlegal.astype(int)
        uni = np.array([[1.0 / n[i] if c < n[i] else 0.0 for c in range(4)] for i in range(len(n))])
        sigma = np.where(s == 0, uni, pos / np.where(s == 0, 1, s))
        self.R += self.delta[:, :4]
        self.S += self.delta[:, 4:5] * sigma
        self.delta[:] = 0
        self.it += 1


def _worker(rank, world, port, batch, iters, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from scopa_amd.distributed import ShardedMCCFR
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = OracleEngine(seed=31337)
    buf = torch.from_numpy(eng.delta)  # aliases the delta buffer, as the bound torch tensor does on the GPU

    def all_reduce():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    ShardedMCCFR(eng, rank, world, all_reduce).run(batch, iters)
    np.savez(out.format(rank=rank), R=eng.R, S=eng.S)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from scopa_amd.distributed import shard_range
    for total in (0, 1, 7, 4096, 262144, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(nb for _, nb in spans) == total
            assert all(spans[r][0] + spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            assert max(nb for _, nb in spans) - min(nb for _, nb in spans) <= 1


def test_two_ranks_reproduce_one_rank(tmp_path):
    import torch.multiprocessing as mp
    batch, iters, world = 37, 4, 2   # odd batch: uneven shards
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank{rank}.npz")
    mp.spawn(_worker, args=(world, port, batch, iters, out), nprocs=world, join=True)
    single = OracleEngine(seed=31337)
    from scopa_amd.distributed import ShardedMCCFR
    ShardedMCCFR(single, 0, 1).run(batch, iters)
    r0, r1 = np.load(out.format(rank=0)), np.load(out.format(rank=1))
    assert np.array_equal(r0["R"], r1["R"]) and np.array_equal(r0["S"], r1["S"])   # replicas stay bit-identical
    np.testing.assert_allclose(r0["R"], single.R, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r0["S"], single.S, rtol=1e-12, atol=1e-12)
    # and the host-side apply of the test double is the oracle's own batched iteration
    ref = OracleEngine(seed=31337)
    ref.t.mccfr_batched(ref.R, ref.S, 31337, 0, iters, batch)
    np.testing.assert_allclose(single.R, ref.R, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(single.S, ref.S, rtol=1e-12, atol=1e-12)


def test_eight_ranks_reproduce_one_rank(tmp_path):
    """BASELINE configs[2]'s rank count: EIGHT gloo processes (one per host core) through the same driver -- shard_range x 8,
    one all-reduce of the [738][5] delta per iteration, every rank applying the sum -- reproduce the one-rank tables; the eight
    replicas end bit-identical.  (Batch 8 x 37 + 3: uneven shards; the GPU box admits at most five ranks on its one device,
    tests/test_gpu_bench.py, so the eight-rank control flow is exercised here.)"""
    import torch.multiprocessing as mp
    batch, iters, world = 8 * 37 + 3, 3, 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank{rank}.npz")
    mp.spawn(_worker, args=(world, port, batch, iters, out), nprocs=world, join=True)
    single = OracleEngine(seed=31337)
    from scopa_amd.distributed import ShardedMCCFR, shard_range
    assert [shard_range(batch, r, world)[1] for r in range(world)] == [38, 38, 38, 37, 37, 37, 37, 37]
    ShardedMCCFR(single, 0, 1).run(batch, iters)
    reps = [np.load(out.format(rank=r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(reps[0]["R"], reps[r]["R"]) and np.array_equal(reps[0]["S"], reps[r]["S"]), r
    np.testing.assert_allclose(reps[0]["R"], single.R, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(reps[0]["S"], single.S, rtol=1e-12, atol=1e-12)


def test_light_form_is_guarded_by_periodic_replica_checks():
    """ShardedMCCFR.run under the LIGHT protocol form of the peer exchange calls the replica check every check_every iterations
    (and never under the fenced form or the collective): a divergence raises instead of corrupting the tables silently."""
    from scopa_amd.distributed import ShardedMCCFR

    class Eng:
        def __init__(self, form):
            self.exchange_form, self.calls, self.checks = form, [], 0
        def mccfr_iterate_sharded(self, b0, nb, n):
            self.calls.append(n)
        def replica_check(self):
            self.checks += 1

    e = Eng("light")
    ShardedMCCFR(e, 0, 2, fused_exchange=True, check_every=100).run(64, 250)
    assert e.calls == [100, 100, 50] and e.checks == 2
    e = Eng("fenced")
    ShardedMCCFR(e, 0, 2, fused_exchange=True, check_every=100).run(64, 250)
    assert e.calls == [250] and e.checks == 0
    bad = Eng("light")
    bad.replica_check = lambda: (_ for _ in ()).throw(RuntimeError("tables differ"))
    with pytest.raises(RuntimeError):
        ShardedMCCFR(bad, 0, 2, fused_exchange=True, check_every=10).run(64, 25)


# ---- SDCFR data parallelism: gradient averaging keeps replicas identical (BASELINE configs[4]) -----------------------
def _grad_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from scopa_amd.distributed import allreduce_gradients, broadcast_parameters
    from scopa_amd.algorithms.deep_cfr.nets import FlexibleNet
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank ...
    net = FlexibleNet(input_shape=(34,), output_dim=16, mlp_hidden=[128, 64])
    broadcast_parameters(net, lambda t: dist.broadcast(t, src=0))   # ... made identical
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.rand(64, 34, generator=g), torch.rand(64, 16, generator=g)   # the global batch, same on both ranks
    lo, hi = rank * 32, rank * 32 + 32                                          # each rank owns half of it
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(X[lo:hi]), Y[lo:hi]).backward()
        allreduce_gradients(net.parameters(), world, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        opt.step()
    torch.save(net.state_dict(), out.format(rank=rank))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_equals_full_batch_training(tmp_path):
    import torch
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from scopa_amd.algorithms.deep_cfr.nets import FlexibleNet
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "net{rank}.pt")
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = torch.load(out.format(rank=0), weights_only=True), torch.load(out.format(rank=1), weights_only=True)
    assert all(torch.equal(a[k], b[k]) for k in a)      # replicas identical after 3 steps
    # and equal to single-process training on the whole batch (mean of two half-batch MSE grads = full-batch grad)
    torch.manual_seed(100)
    net = FlexibleNet(input_shape=(34,), output_dim=16, mlp_hidden=[128, 64])
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.rand(64, 34, generator=g), torch.rand(64, 16, generator=g)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(X), Y).backward()
        opt.step()
    for k, v in net.state_dict().items():
        assert torch.allclose(v, a[k], atol=1e-6), k
