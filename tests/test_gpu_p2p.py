"""The N > 1 exchange step of the batched-MCCFR path done by the library itself: one-shot all-reduce of the delta buffer
through peer-mapped (hipIpc) fine-grained memory (scopa_p2p_*, scopa_amd/csrc/scopa_p2p.hip).  The test box has ONE GPU, so
the ranks are separate processes sharing it: that exercises the IPC plumbing, the flag/parity protocol, the rank-ordered sum
and the bounded waits; the xGMI hop itself can only be exercised by the driver's multi-GPU run, where bench.py validates the
exchange against RCCL before using it (scopa_amd.distributed.connect_peer_exchange)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_self_exchange(ctx, sl):
    ctx.set_deal(sl.deal_py_seed(42))
    h = ctx.p2p_create(0, 1)
    ctx.p2p_connect(h.reshape(1, 64))
    rng = np.random.RandomState(0)
    for k in range(5):                                   # both parities, several sequence numbers
        x = rng.standard_normal((ctx.n_infosets, 5))
        ctx.mccfr_delta_set(x)
        ctx.p2p_allreduce_delta()
        assert np.array_equal(ctx.mccfr_delta_get(), x)
    assert ctx.p2p_status() == (0, 5)
    ctx.p2p_destroy()
    with pytest.raises(sl.ScopaError):
        ctx.p2p_allreduce_delta()


def test_odd_sized_delta_is_exchanged_whole(ctx, sl):
    # a deal whose infoset count is odd: n*5 float64 is not a whole number of 16-byte pieces
    for seed in range(1, 200):
        if ctx.set_deal(sl.deal_py_seed(seed)) % 2 == 1:
            break
    assert ctx.n_infosets % 2 == 1
    h = ctx.p2p_create(0, 1)
    ctx.p2p_connect(h.reshape(1, 64))
    x = np.random.RandomState(1).standard_normal((ctx.n_infosets, 5))
    ctx.mccfr_delta_set(x)
    ctx.p2p_allreduce_delta()
    assert np.array_equal(ctx.mccfr_delta_get(), x)
    ctx.p2p_destroy()
    ctx.set_deal(sl.deal_py_seed(42))


def _run_ranks(tmp_path, world, batch_total, iters, mode, port, form="auto"):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SCOPA_TEST_P2P_FORM=form)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "p2p_worker.py"), str(r), str(world), str(port),
                               str(tmp_path), str(batch_total), str(iters), mode], env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert all(bool(r["ok"]) for r in res), [str(r["why"]) for r in res]
    assert [int(r["timeouts"]) for r in res] == [0] * world
    assert [int(r["exchanges"]) for r in res] == [32 + iters] * world          # 32 validation rounds + the run
    assert len({str(r["form"]) for r in res}) == 1 and str(res[0]["form"]) in (("light", "fenced") if form == "auto" else (form,))
    for r in res[1:]:                                    # rank-ordered sums: replicas are bit-identical
        assert np.array_equal(r["R"], res[0]["R"]) and np.array_equal(r["S"], res[0]["S"])
    assert sum(int(r["visits"]) for r in res) == 463 * batch_total * iters
    return res


def _single_process(ctx, sl, batch_total, iters):
    ctx.set_deal(sl.deal_py_seed(42))
    ctx.mccfr_seed(77)
    ctx.mccfr_iterate(batch_total, iters)
    R, S, _ = ctx.tables_get()
    return R, S


def test_two_ranks_free_running_fused_and_split(ctx, sl, tmp_path):
    """Two processes sharing the GPU, no host synchronisation inside the run: 3 fused iterations (exchange inside the
    reduce+apply kernel, in-library loop) then 3 split ones (traverse+reduce, stand-alone exchange, apply)."""
    batch_total, iters = 1001, 6                         # 1001 = 501 + 500 traversal ids
    res = _run_ranks(tmp_path, 2, batch_total, iters, "free", 29611)
    # the same global traversal ids in one process: equal up to the summation order of the partial deltas
    R, S = _single_process(ctx, sl, batch_total, iters)
    assert np.allclose(res[0]["R"], R, rtol=1e-12, atol=1e-12) and np.allclose(res[0]["S"], S, rtol=1e-12, atol=1e-12)


def test_two_ranks_free_running_fenced_form(ctx, sl, tmp_path):
    """The library's default protocol form (plain accesses + system-scope release/acquire fences), forced."""
    batch_total, iters = 515, 4
    res = _run_ranks(tmp_path, 2, batch_total, iters, "free", 29614, form="fenced")
    R, S = _single_process(ctx, sl, batch_total, iters)
    assert np.allclose(res[0]["R"], R, rtol=1e-12, atol=1e-12) and np.allclose(res[0]["S"], S, rtol=1e-12, atol=1e-12)


def test_absent_peer_is_reported_not_applied_silently(tmp_path):
    """A peer that never answers: the bounded waits give up (no hang) and scopa_p2p_allreduce_delta / scopa_mccfr_iterate_sharded
    / ShardedMCCFR.run all fail with SCOPA_ETIMEOUT, on that call and on every later one."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "p2p_worker.py"), str(r), "2", "29615", str(tmp_path), "64", "1", "timeout"],
                              env=env) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=180) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    r0 = np.load(tmp_path / "rank0.npz")
    assert bool(r0["ok"]), str(r0["why"])
    assert list(r0["statuses"]) == [-7, -7, -7] and int(r0["timeouts"]) > 0


def test_three_ranks_lockstep_rank_ordered_sum(ctx, sl, tmp_path):
    batch_total, iters = 1000, 4                         # 1000 = 334 + 333 + 333 traversal ids
    res = _run_ranks(tmp_path, 3, batch_total, iters, "lockstep", 29612)
    R, S = _single_process(ctx, sl, batch_total, iters)
    assert np.allclose(res[0]["R"], R, rtol=1e-12, atol=1e-12) and np.allclose(res[0]["S"], S, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("mode", ["fail1", "failconnect"])
def test_setup_failure_on_one_rank_is_agreed_by_all(tmp_path, mode):
    """The fallback decision (peer exchange vs torch.distributed all-reduce) must be the same on every rank."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "p2p_worker.py"), str(r), "2", "29613", str(tmp_path), "64", "1", mode],
                              env=env) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=120) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    assert [bool(r["ok"]) for r in res] == [False, False], [str(r["why"]) for r in res]
