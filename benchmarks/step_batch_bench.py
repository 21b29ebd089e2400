#!/usr/bin/env python3
"""scopa_step_batch (k_step_batch): n independent MiniScopa games advance one ply in lockstep -- the state engine's
HBM-bound kernel (16 B state in, 1 B action in, 16 B state out per game; mini_scopa_game.py:140-167 per lane).
Games are dealt from distinct seeds on the host for a small pool, replicated on the device, and played with uniformly
random legal cards for all 8 plies; reports games*plies/s and achieved HBM GB/s against the 33 algorithmic bytes per step.

    python benchmarks/step_batch_bench.py --games 67108864
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1 << 26)
    a = ap.parse_args()
    import torch
    from scopa_amd import _lib
    ctx = _lib.Context(0)
    pool = np.zeros(4096, _lib.STATE_DTYPE)
    for i in range(pool.size):                                   # MiniScopaGame.reset for 4096 different deals
        p = _lib.deal_py_seed(i)
        pool[i]["hand"] = (int(p[0]) | int(p[1]) << 4 | int(p[2]) << 8 | int(p[3]) << 12, int(p[4]) | int(p[5]) << 4 | int(p[6]) << 8 | int(p[7]) << 12)
        pool[i]["nh"] = (4, 4)
    dev = torch.device("cuda:0")
    base = torch.from_numpy(pool.view(np.uint8).reshape(pool.size, 16)).to(dev)
    reps = (a.games + pool.size - 1) // pool.size
    states = base.repeat(reps, 1)[:a.games].contiguous()
    g = torch.Generator(device=dev); g.manual_seed(0)
    times = []
    for ply in range(8):
        # a uniformly random LEGAL card: the ply's mover holds nh cards, ordered nibbles in hand[mover]
        mover = ply & 1
        hand = states[:, 2 * mover].to(torch.int32) | (states[:, 2 * mover + 1].to(torch.int32) << 8)
        nh = states[:, 8 + mover].to(torch.int32).clamp(min=1)
        k = (torch.rand(a.games, device=dev, generator=g) * nh).to(torch.int32).clamp(max=3)
        actions = ((hand >> (4 * k)) & 15).to(torch.uint8).contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.step_batch(states.data_ptr(), actions.data_ptr(), a.games)
        ctx.synchronize()
        times.append(time.perf_counter() - t0)
    st = states.cpu().numpy().view(_lib.STATE_DTYPE).reshape(-1)
    assert (st["step"] == 8).all() and (st["nh"] == 0).all()                      # every game ran to its end
    assert ((st["ncap"].sum(axis=1) + st["nt"]) == 8).all()                       # the 8 cards played are captured or still on the table
    best = min(times[1:])
    print(json.dumps({"kernel": "k_step_batch", "games": a.games, "plies": 8, "seconds_per_ply_best": best, "seconds_per_ply_all": times,
                      "game_steps_per_s": a.games / best, "algorithmic_bytes_per_step": 33, "achieved_GBps": a.games * 33 / best / 1e9,
                      "hbm_peak_GBps": 8000.0, "frac": a.games * 33 / best / 1e9 / 8000.0}))


if __name__ == "__main__":
    main()
