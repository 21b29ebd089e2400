#!/usr/bin/env python3
"""Where an AdvantageNetwork.train() call (graph mode) spends its time: host work to draw the index batches, graph replays, the final sync.
    python tests/tools/sdcfr_train_breakdown.py [batch] [epochs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scopa_amd.algorithms.deep_cfr import DeepCFR
from scopa_amd.envs import load_game
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = DeepCFR(load_game("mini_scopa"), device="cuda:0", batch=B, graph_training=True)
for lean in (True, False):
    for a in d.advantage_nets:
        a.lean_step = lean; a._graphs = {}
    for _ in range(3):
        for p in range(2):
            d._traverse_batch(p, B)
            with torch.cuda.stream(d._stream):
                d.advantage_nets[p].train(epochs=E)
    a = d.advantage_nets[0]
    n = len(a.buffer)
    with torch.cuda.stream(d._stream):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            a.train(epochs=E)
        torch.cuda.synchronize(); t_train = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            a._rng.seed(42); a._rng.shuffle(list(range(16)))
            rows_all = a._sample_rows(n, 128, E)
        torch.cuda.synchronize(); t_sample = (time.perf_counter() - t0) / 20
        g, rows, losses = a._graphs[(128, E)]
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize(); t_replay = (time.perf_counter() - t0) / 20 / E
        t0 = time.perf_counter()
        for _ in range(20):
            rows.copy_(rows_all); g.replay()
        torch.cuda.synchronize(); t_replay_copy = (time.perf_counter() - t0) / 20 / E
    print(f"lean={lean}: train({E} epochs) {1e6 * t_train:.0f} us; drawing + uploading the index batches {1e6 * t_sample:.0f} us; the graph of all epochs, per epoch {1e6 * t_replay:.0f} us; with its index copy {1e6 * t_replay_copy:.0f} us")

# ---- the opt-in HIP backend: the same train() call in 2 x epochs hand-written launches
d2 = DeepCFR(load_game("mini_scopa"), device="cuda:0", batch=B, train_backend="hip")
for _ in range(3):
    for p in range(2):
        d2._traverse_batch(p, B)
        with torch.cuda.stream(d2._stream):
            d2.advantage_nets[p].train(epochs=E)
a = d2.advantage_nets[0]
with torch.cuda.stream(d2._stream):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        a.train(epochs=E)
    torch.cuda.synchronize(); t_train = (time.perf_counter() - t0) / 20
    for bs in (128, 4096):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.train(batch_size=bs, epochs=E); torch.cuda.synchronize()
        rows_all = a._sample_rows(len(a.buffer), bs, E).contiguous(); state, loss = a._hip; ptrs = tuple(p.data_ptr() for p in a.net.parameters())
        torch.cuda.synchronize(); e0.record(d2._stream)
        for rep in range(10):
            for e in range(E):
                a._hip_step += 1
                a._ctx.sdcfr_train_step(rows_all[e].data_ptr(), bs, a.buffer.feat.data_ptr(), a.buffer.regret.data_ptr(), a.buffer.mask_ptr[0], a.buffer.capacity, ptrs, state.data_ptr(), a._hip_step, 5e-4, loss.data_ptr())
        e1.record(d2._stream); e1.synchronize()
        print(f"hip backend: optimiser step on {bs} rows {1e3 * e0.elapsed_time(e1) / (10 * E):.1f} us (events around {10 * E} steps)")
print(f"hip backend: train({E} epochs) {1e6 * t_train:.0f} us")
