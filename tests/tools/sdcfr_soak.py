#!/usr/bin/env python3
"""One long run of the Single Deep CFR loop (a soak, run once per round on the GPU box): N iterations of DeepCFR.train at 4096
traversals per player with the graphed optimiser step, in chunks; after every chunk the exact visit count, the ring's write position,
the shape of the rows the last launch wrote (one-hot features, masks with 1..4 legal actions, regrets within [-1, 1] with one common value
in the illegal slots, as the reference stores them), finite nets, and the average policy's reward against uniform random over 4096 episodes.
    python tests/tools/sdcfr_soak.py [iterations]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import contextlib
import numpy as np
import torch
from scopa_amd.algorithms.deep_cfr import DeepCFR
from scopa_amd.envs import load_game

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
B = 4096
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    d = DeepCFR(load_game("mini_scopa"), device="cuda:0", batch=B, graph_training=True)
ctx = d._engine.ctx
v0, done, pts, t0 = ctx.sdcfr_visits(), 0, [], time.perf_counter()
chunks = [10, 90, 400] + [500] * max(0, (N - 500) // 500)
for chunk in chunks:
    d.train(iterations=chunk, advantage_epochs=10, eval_freq=10 ** 9)
    done += chunk
    assert ctx.sdcfr_visits() - v0 == (105 + 82) * B * done, (ctx.sdcfr_visits() - v0, done)
    for p, a in enumerate(d.advantage_nets):
        mem = a.buffer
        assert mem.write_base == (41 * B * done) % mem.capacity and len(mem) == min(mem.capacity, 41 * B * done)
        last = (mem.write_base - 41 * B + torch.arange(41 * B, device="cuda:0")) % mem.capacity       # the rows of the last launch
        f, r, m = mem.feat[last], mem.regret[last], mem.mask[last]
        assert torch.isfinite(r).all() and bool(((m == 0) | (m == 1)).all()) and bool(((f == 0) | (f == 1)).all())
        nl = m.sum(1)
        assert bool(((nl >= 1) & (nl <= 4)).all()) and bool((r.abs() <= 1).all())
        ill = torch.where(m == 0, r, r.new_full((), float('nan')))                           # the reference stores -value/den in every illegal slot: one value per row
        assert bool((torch.nan_to_num(ill, nan=-9).amax(1) == torch.nan_to_num(ill, nan=9).amin(1)).all())
        assert bool((f[:, :16].sum(1) == nl).all()) and bool((f[:, 32] == 1).all()) and bool((f[:, 33] == 0).all())   # hand size = legal actions; the mover's own row
        assert all(bool(torch.isfinite(t).all()) for t in a.net.state_dict().values())
    reward, scopas = d.evaluate_vs_random(num_episodes=4096)
    pts.append({"iterations": done, "reward_vs_random": reward, "scopas_trained_vs_random": scopas, "loss": [d.training_history["losses"][p][-1] for p in (0, 1)],
                "seconds": time.perf_counter() - t0})
    print(pts[-1], file=sys.stderr, flush=True)
print(json.dumps({"batch_per_player": B, "graph_training": True, "points": pts, "visits": ctx.sdcfr_visits() - v0,
                  "rows_written": 2 * 41 * B * done, "iterations_per_s_incl_checks": done / (time.perf_counter() - t0)}))
