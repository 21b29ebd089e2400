#!/usr/bin/env python3
"""SDCFR throughput (BASELINE configs[3]/[4]): B concurrent traversals per player per iteration through the HIP
traversal kernels + PyTorch-ROCm advantage MLP; reports decision-node visits/s of the traversal and the cost of the
training step beside it.  N>1: torch.distributed.run, one rank per GPU (sharded traversals + gradient all-reduce).

    python benchmarks/sdcfr_bench.py --batch 4096 --iters 20
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--ply-by-ply", action="store_true", help="use the level-synchronous multi-launch traversal instead of the fused kernel")
    ap.add_argument("--graph", action="store_true", help="replay the optimiser step as a HIP graph")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.deep_cfr import DeepCFR
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.manual_seed(0)
    d = DeepCFR(load_game("mini_scopa"), device=f"cuda:{local}", batch=args.batch, rank=rank, world=world, graph_training=args.graph, fused_traversal=not args.ply_by_ply)
    ctx = d._engine.ctx

    def one_iter(train=True):
        tt = 0.0
        for p in range(2):
            a = time.perf_counter()
            d._traverse_batch(p, args.batch)
            tt += time.perf_counter() - a
            if train:
                with torch.cuda.stream(d._stream):
                    d.advantage_nets[p].train(epochs=args.epochs)
                d._stream.synchronize()
        d._iteration += 1
        return tt

    for _ in range(args.warmup):
        one_iter()
    torch.cuda.synchronize()
    v0 = ctx.sdcfr_visits()
    t0 = time.perf_counter()
    trav_s = sum(one_iter() for _ in range(args.iters))
    torch.cuda.synchronize()
    total_s = time.perf_counter() - t0
    visits = ctx.sdcfr_visits() - v0
    assert visits == (105 + 82) * args.batch * args.iters
    if rank == 0:
        print(json.dumps({"workload": f"SDCFR, {args.batch} traversals/player/iteration/GPU, MLP 34-128-64-16 f32, {args.epochs} Adam epochs x batch 128",
                          "n_gpus": world, "iterations": args.iters, "decision_visits_per_gpu": visits,
                          "traversal_visits_per_s_per_gpu": visits / trav_s, "traversal_ms_per_iteration": 1e3 * trav_s / args.iters,
                          "iteration_ms_incl_training": 1e3 * total_s / args.iters,
                          "memory_rows_per_iteration": 2 * 41 * args.batch,
                          "reference_python_visits_per_s": "930-2900 (BASELINE.md §2, 1 Xeon core, batch-1 forward per node)"}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
