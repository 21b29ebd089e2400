#!/usr/bin/env python3
"""One rank of the peer-memory exchange test (tests/test_gpu_p2p.py): all ranks share GPU 0 (the test box has one), the
process group is gloo (RCCL refuses two ranks on one device), the exchange itself is the library's hipIpc path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    batch_total, iters, mode = int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    import torch
    import torch.distributed as dist
    from scopa_amd import _lib
    from scopa_amd.distributed import ShardedMCCFR, connect_peer_exchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ctx = _lib.Context(0)
    ctx.set_deal(_lib.deal_py_seed(42))
    ctx.mccfr_seed(77)
    if mode == "fail1" and rank == 1:            # one rank cannot set the exchange up: every rank must come back with ok == False, nobody hangs
        def refuse(*a, **k):
            raise _lib.ScopaError(-3, "scopa_p2p_create (simulated failure)")
        ctx.p2p_create = refuse
    if mode == "failconnect" and rank == 0:
        def refuse2(*a, **k):
            raise _lib.ScopaError(-3, "scopa_p2p_connect (simulated failure)")
        ctx.p2p_connect = refuse2
    form = os.environ.get("SCOPA_TEST_P2P_FORM", "auto")
    ok, why = connect_peer_exchange(ctx, rank, world, torch.device("cuda:0"), rounds=32, form=form)
    res = dict(ok=ok, why=why, form=str(getattr(ctx, "exchange_form", None)))
    if ok and mode == "timeout":
        # rank 1 never takes part in the next exchange: rank 0's bounded waits give up, and the library must SAY so -- on the call
        # that timed out and on every later one (the tables have had partial sums applied)
        statuses = []
        if rank == 0:
            ctx.p2p_set_budget(0.05)
            for call in (ctx.p2p_allreduce_delta, lambda: ctx.mccfr_iterate_sharded(0, 64, 1), lambda: ShardedMCCFR(ctx, 0, world, fused_exchange=True).run(128, 2)):
                try:
                    call()
                    statuses.append(0)
                except _lib.ScopaError as e:
                    statuses.append(e.status)
        res.update(statuses=np.array(statuses), timeouts=ctx.p2p_status()[0])
    elif ok:
        if mode == "free":
            # first half: the fused form (exchange inside the reduce+apply kernel, in-library loop); second half: the split form
            # (traverse+reduce, stand-alone row exchange, apply) -- both are product paths over the same inbox protocol
            ShardedMCCFR(ctx, rank, world, fused_exchange=True).run(batch_total, iters // 2)
            ShardedMCCFR(ctx, rank, world, ctx.p2p_allreduce_delta).run(batch_total, iters - iters // 2)
        else:
            # "lockstep": the split form with a host barrier before every exchange.  The ranks SHARE one GPU here and the traversal
            # kernel needs a whole CU's LDS, so a rank spinning in an exchange while another still has its traversal queued could
            # starve it (with one process per GPU -- the deployment -- a GPU only ever runs its own stream)
            from scopa_amd.distributed import shard_range
            b0, nb = shard_range(batch_total, rank, world)
            for _ in range(iters):
                ctx.mccfr_traverse(ctx.mccfr_iteration(), b0, nb)
                ctx.synchronize()
                dist.barrier()
                ctx.p2p_allreduce_delta()
                ctx.mccfr_apply()
        timeouts, exchanges = ctx.p2p_status()
        R, S, _ = ctx.tables_get()
        res.update(timeouts=timeouts, exchanges=exchanges, R=R, S=S, visits=ctx.counters()[0])
    np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
