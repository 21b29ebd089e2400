#!/usr/bin/env python3
"""The three batched state engines as north_star describes them -- independent games held as packed states in HBM and advanced one ply
per launch, one lane per game: load the state, apply the move (legal-move check, capture search, scopa, scoring at the end), store it.

    engine                         state    algorithmic bytes per game-step
    MiniScopa   k_step_batch       16 B     16 in + 1 action + 16 out = 33      (mini_scopa_game.py:140-167)
    Team TPI    k_team_step_batch  40 B     40 + 1 + 40 = 81                    (team_mini_scopa_game.py, openspiel_team_mini_scopa.py)
    FullScopa   k_full_step_batch  64 B     64 + 1 + 64 = 129 (+ the game's 40-byte deck at a re-deal, cache-resident) (full_scopa_game.py)

Games are dealt on the host for a pool of seeds, replicated on the device and played to the end with uniformly random LEGAL cards (drawn
with torch from the packed hands); every ply's launch is timed between stream synchronisations (0.3-3 ms per launch).  Reports game-steps/s and achieved
HBM GB/s against the algorithmic bytes, per engine: the best ply, and the whole game.

    python benchmarks/state_engines_bench.py [--mini 67108864 --team 33554432 --full 16777216]
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBPS = 8000.0


def timed(ctx, torch, fn, stream=None):
    """one launch on the context's stream.  stream (a torch stream the context was created on): HIP events recorded on that stream around the
    launch -- the kernel's own duration; without it the host clock between synchronisations (launches here take 0.1-3 ms; the bracket costs ~10 us)"""
    import time
    torch.cuda.synchronize(); ctx.synchronize()
    if stream is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream)
        e1.synchronize()
        return 1e-3 * e0.elapsed_time(e1)
    t0 = time.perf_counter(); fn(); ctx.synchronize()
    return time.perf_counter() - t0


def report(name, n, bytes_step, times, extra):
    best = min(times)
    med = sorted(times)[len(times) // 2]
    total = sum(min(t, 3.0 * med) for t in times)          # a launch that hits a clock event is counted at 3 x the median ply (each kernel's first launch, which loads its code, is made before the plies)

    return {"kernel": name, "games": n, "plies": len(times), "algorithmic_bytes_per_step": bytes_step,
            "best_ply": {"seconds": best, "game_steps_per_s": n / best, "achieved_GBps": n * bytes_step / best / 1e9, "frac": n * bytes_step / best / 1e9 / HBM_PEAK_GBPS},
            "median_ply": {"seconds": med, "game_steps_per_s": n / med, "achieved_GBps": n * bytes_step / med / 1e9, "frac": n * bytes_step / med / 1e9 / HBM_PEAK_GBPS},
            "whole_game": {"seconds": total, "game_steps_per_s": n * len(times) / total, "achieved_GBps": n * len(times) * bytes_step / total / 1e9,
                           "frac": n * len(times) * bytes_step / total / 1e9 / HBM_PEAK_GBPS},
            "seconds_per_ply": times, **extra}


def measure(ctx, n_mini=1 << 26, n_team=1 << 25, n_full=1 << 24, n_pool=4096, stream=None, device=0, warm_up=True):
    """The three engines on context `ctx`; stream: the torch stream the context launches on (event timing), or None (host clock)."""
    import types
    import torch
    from scopa_amd import _lib
    a = types.SimpleNamespace(mini=n_mini, team=n_team, full=n_full, pool=n_pool)
    dev = torch.device(f"cuda:{device}")
    g = torch.Generator(device=dev); g.manual_seed(0)
    out = {"hbm_peak_GBps": HBM_PEAK_GBPS,
           "timing": "per launch; " + ("HIP events on the kernels' stream" if stream is not None else "host clock between stream synchronisations")}

    def rnd_k(nh):       # uniform index below nh (at least 1)
        return (torch.rand(nh.numel(), device=dev, generator=g) * nh.clamp(min=1)).to(torch.int32).clamp(max=3).minimum(nh.clamp(min=1) - 1)

    # ---- MiniScopa ------------------------------------------------------------------------------------------------------
    if a.mini:
        pool = np.zeros(a.pool, _lib.STATE_DTYPE)
        for i in range(pool.size):
            p = _lib.deal_py_seed(i)
            pool[i]["hand"] = (int(p[0]) | int(p[1]) << 4 | int(p[2]) << 8 | int(p[3]) << 12, int(p[4]) | int(p[5]) << 4 | int(p[6]) << 8 | int(p[7]) << 12)
            pool[i]["nh"] = (4, 4)
        base = torch.from_numpy(pool.view(np.uint8).reshape(pool.size, 16)).to(dev)
        st = base.repeat((a.mini + pool.size - 1) // pool.size, 1)[:a.mini].contiguous()
        if warm_up:                                                    # the kernel's first launch loads its code object: not a ply's time
            warm = base.clone(); wact = torch.zeros(pool.size, dtype=torch.uint8, device=dev)
            ctx.step_batch(warm.data_ptr(), wact.data_ptr(), pool.size); ctx.synchronize()
        times = []
        for ply in range(8):
            mover = ply & 1
            hand = st[:, 2 * mover].to(torch.int32) | (st[:, 2 * mover + 1].to(torch.int32) << 8)
            k = rnd_k(st[:, 8 + mover].to(torch.int32))
            act = ((hand >> (4 * k)) & 15).to(torch.uint8).contiguous()
            del hand, k
            times.append(timed(ctx, torch, lambda: ctx.step_batch(st.data_ptr(), act.data_ptr(), a.mini), stream))
        h = st[:1 << 20].cpu().numpy().view(_lib.STATE_DTYPE).reshape(-1)
        assert (h["step"] == 8).all() and (h["nh"] == 0).all() and ((h["ncap"].sum(axis=1) + h["nt"]) == 8).all()
        out["mini"] = report("k_step_batch", a.mini, 33, times, {})
        del st, act
        torch.cuda.empty_cache()

    # ---- Team MiniScopa TPI: 4 seats x 4 cards, 16 plies ----------------------------------------------------------------
    if a.team:
        pool = np.zeros(a.pool, _lib.TEAM_STATE_DTYPE)
        for i in range(pool.size):
            pool[i] = _lib.TeamState(seed=i).s[0]
        base = torch.from_numpy(pool.view(np.uint8).reshape(pool.size, 40)).to(dev)
        st = base.repeat((a.team + pool.size - 1) // pool.size, 1)[:a.team].contiguous()
        if warm_up:
            warm = base.clone(); wact = torch.zeros(pool.size, dtype=torch.uint8, device=dev)
            ctx.team_step_batch(warm.data_ptr(), wact.data_ptr(), pool.size); ctx.synchronize()
        times = []
        for ply in range(16):
            seat = ply & 3                                             # hand[4] u16 at bytes 12..19, nh[4] at bytes 28..31
            hand = st[:, 12 + 2 * seat].to(torch.int32) | (st[:, 13 + 2 * seat].to(torch.int32) << 8)
            k = rnd_k(st[:, 28 + seat].to(torch.int32))
            act = ((hand >> (4 * k)) & 15).to(torch.uint8).contiguous()
            del hand, k
            times.append(timed(ctx, torch, lambda: ctx.team_step_batch(st.data_ptr(), act.data_ptr(), a.team), stream))
        h = st[:1 << 20].cpu().numpy().view(_lib.TEAM_STATE_DTYPE).reshape(-1)
        assert (h["step"] == 16).all() and (h["nh"] == 0).all() and (h["flags"] & 1).all()
        out["team"] = report("k_team_step_batch", a.team, 81, times, {})
        del st, act
        torch.cuda.empty_cache()

    # ---- FullScopa: 40 cards, 3-card hands re-dealt six times, 36 plies ---------------------------------------------------
    if a.full:
        fpool = min(a.pool, 1024)                                      # a FullScopa deal costs 0.6 ms on the host
        pool = np.zeros(fpool, _lib.FULL_STATE_DTYPE)
        decks = np.zeros((fpool, 40), np.uint8)
        for i in range(pool.size):
            decks[i] = _lib.full_deal_py_seed(i)
            pool[i] = _lib.FullState(deck=decks[i], game=i).s[0]
        d_decks = torch.from_numpy(decks).to(dev)
        base = torch.from_numpy(pool.view(np.uint8).reshape(pool.size, 64)).to(dev)
        st = base.repeat((a.full + pool.size - 1) // pool.size, 1)[:a.full].contiguous()
        if warm_up:
            warm = base.clone(); wact = torch.zeros(fpool, dtype=torch.uint8, device=dev)
            ctx.full_step_batch(warm.data_ptr(), wact.data_ptr(), d_decks.data_ptr(), fpool); ctx.synchronize()
        times, plies = [], 0
        for ply in range(40):
            if bool((st[:1 << 16, 54] != 0).all()):                   # terminal flag of a sample: every game has the same length
                break
            p = ply & 1                                                # hand[2] u32 (3 six-bit slots) at bytes 32..39, nh[2] at bytes 44, 45
            hand = st[:, 32 + 4 * p].to(torch.int32) | (st[:, 33 + 4 * p].to(torch.int32) << 8) | (st[:, 34 + 4 * p].to(torch.int32) << 16)
            nh = st[:, 44 + p].to(torch.int32)
            k = rnd_k(nh)
            act = torch.where(nh > 0, (hand >> (6 * k)) & 63, torch.zeros_like(hand)).to(torch.uint8).contiguous()
            del hand, k, nh
            times.append(timed(ctx, torch, lambda: ctx.full_step_batch(st.data_ptr(), act.data_ptr(), d_decks.data_ptr(), a.full), stream))
            plies += 1
        h = st[:1 << 20].cpu().numpy().view(_lib.FULL_STATE_DTYPE).reshape(-1)
        assert (h["terminal"] == 1).all() and (h["flags"] == 0).all()
        out["full"] = report("k_full_step_batch", a.full, 129, times, {"game_length_plies": plies})
        del st, act
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mini", type=int, default=1 << 26)
    ap.add_argument("--team", type=int, default=1 << 25)
    ap.add_argument("--full", type=int, default=1 << 24)
    ap.add_argument("--pool", type=int, default=4096)
    ap.add_argument("--host-clock", action="store_true", help="time launches with the host clock between synchronisations instead of HIP events")
    ap.add_argument("--no-warm-up", action="store_true", help="no small first launch per kernel (under a profiler, whose per-kernel averages would mix it in)")
    a = ap.parse_args()
    import torch
    from scopa_amd import _lib
    stream = None if a.host_clock else torch.cuda.Stream()
    ctx = _lib.Context(0, stream=stream.cuda_stream if stream is not None else None)
    print(json.dumps(measure(ctx, a.mini, a.team, a.full, a.pool, stream, warm_up=not a.no_warm_up)))


if __name__ == "__main__":
    main()
