#!/usr/bin/env python3
"""Two ranks on ONE GPU (gloo group), split path traverse | all-reduce | apply: after every iteration compare the ranks' tables with each
other and with the one-GPU run of the same global traversal ids.   python tests/tools/split_path_probe.py [iterations] [batch_total]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, iters, batch_total, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scopa_amd import _lib
    from scopa_amd.distributed import ShardedMCCFR, make_gpu_engine
    perm = _lib.deal_py_seed(42)
    ctx, delta, stream, all_reduce = make_gpu_engine(0, perm, seed=0x5C09A, distributed=world > 1, rank=rank, exchange="rccl")
    drv = ShardedMCCFR(ctx, rank, world, all_reduce, fused_exchange=False, always_exchange=True)
    ref = None
    if rank == 0:
        ref = _lib.Context(0); ref.set_deal(perm); ref.mccfr_seed(0x5C09A)
    bad = 0
    stride = int(os.environ.get("PROBE_STRIDE", "1"))   # iterations between two checks (no host synchronisation in between)
    for it in range(0, iters, stride):
        drv.run(batch_total, stride)
        torch.cuda.synchronize()
        R, S, _ = ctx.tables_get()
        h = torch.tensor(list(hashlib.sha256(R.tobytes() + S.tobytes()).digest()[:8]), dtype=torch.uint8)
        allh = [torch.zeros_like(h) for _ in range(world)]
        dist.all_gather(allh, h)
        same = all(bool((x == allh[0]).all()) for x in allh)
        msg = ""
        if rank == 0:
            ref.mccfr_iterate(batch_total, stride)
            R1, S1, _ = ref.tables_get()
            msg = f" vs one GPU: max|dR| {np.abs(R - R1).max():.3e} max|dS| {np.abs(S - S1).max():.3e}"
        if rank == 0 and (not same or it < 3 or it == iters - 1):
            print(f"iteration {it}: replicas {'identical' if same else 'DIFFER'}{msg}", flush=True)
        bad += 0 if same else 1
    if rank == 0:
        print(f"{bad} of {iters} iterations with differing replicas", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    bt = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    mp.spawn(worker, args=(2, iters, bt, 29571), nprocs=2, join=True)
