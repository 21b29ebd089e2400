#!/usr/bin/env python3
"""DeepCFR.train as a caller runs it (traversals, optimiser epochs, strategy snapshots; no evaluation): ms per iteration, both training backends,
and one evaluate_vs_random(50) against the 100 stored snapshots.
    python tests/tools/sdcfr_loop_time.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scopa_amd.algorithms.deep_cfr import DeepCFR
from scopa_amd.envs import load_game
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for backend in ("torch", "hip"):
    d = DeepCFR(load_game("mini_scopa"), device="cuda:0", batch=B, graph_training=True, train_backend=backend)
    d.train(iterations=30, advantage_epochs=5, eval_freq=10 ** 9)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.train(iterations=200, advantage_epochs=5, eval_freq=10 ** 9)
    torch.cuda.synchronize()
    print(f"{backend}: DeepCFR.train {1e3 * (time.perf_counter() - t0) / 200:.3f} ms per iteration (5 epochs, snapshots included, no evaluation); {len(d.strategy_buffers[0].strategies)} snapshots")
    t0 = time.perf_counter(); r = d.evaluate_vs_random(50); torch.cuda.synchronize()
    print(f"   evaluate_vs_random(50): {1e3 * (time.perf_counter() - t0):.2f} ms, reward {r[0]:.3f}")
