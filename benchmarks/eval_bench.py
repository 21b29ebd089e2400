#!/usr/bin/env python3
"""The on-device evaluator (evaluate_agent, vanilla_cfr.py:157-216 / mc_cfr.py:146-206): episodes of "tabular average policy vs
uniform random, seats swapped at half time" advanced in lockstep, one lane per episode, one launch per ply (k_eval_tabular_step: 16-byte state
in / out, 4-byte tree index in / out, 4-byte seat, a 32-byte policy row from cache = 44 algorithmic bytes per episode-ply), or the whole match in one launch
as walks over the deal's tree (k_eval_tabular_match: what evaluate_agent_device runs).  Reports episodes/s of both and achieved GB/s of the eight launches.
    python benchmarks/eval_bench.py [--episodes 16777216]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure_kernels(ctx, pol, n, stream=None, device=0):
    """The eight k_eval_tabular_step launches of n episodes on context `ctx` (a deal must be set; pol: device float64 [n_infosets][4] average policy),
    per sampling form; stream = the torch stream the context launches on (HIP events on it) or None (host clock between synchronisations)."""
    import torch
    dev = f"cuda:{device}"
    states = torch.zeros((n, 4), dtype=torch.int32, device=dev); idx = torch.zeros(n, dtype=torch.int32, device=dev)
    seat = (torch.arange(n, device=dev) >= n / 2).to(torch.int32)
    by_form = {}
    for form in ("float64 divisions per visit", "integer thresholds per infoset (scopa_eval_tabular_prepare)"):
        ctx.eval_init_states(states.data_ptr(), n); idx.zero_()
        torch.cuda.synchronize(); ctx.synchronize()
        if form.startswith("integer"):
            ctx.eval_tabular_prepare(pol.data_ptr())
        times = []
        for ply in range(8):
            torch.cuda.synchronize(); ctx.synchronize()
            call = lambda: ctx.eval_tabular_step(states.data_ptr(), idx.data_ptr(), n, ply, 0 if form.startswith("integer") else pol.data_ptr(), seat.data_ptr(), 16)
            if stream is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream); call(); e1.record(stream); e1.synchronize()
                times.append(1e-3 * e0.elapsed_time(e1))
            else:
                t0 = time.perf_counter(); call(); ctx.synchronize(); times.append(time.perf_counter() - t0)
        by_form[form] = {"seconds_8_launches": sum(times), "episodes_per_s": n / sum(times), "achieved_GBps": 8 * n * 44 / sum(times) / 1e9,
                         "frac_of_hbm_peak": 8 * n * 44 / sum(times) / 1e9 / 8000.0, "seconds_per_ply": times}
    # the one-launch form (scopa_eval_tabular_match): synchronous, so the host clock around the call = launch + kernel + the 80-byte statistics copy
    ctx.eval_tabular_prepare(pol.data_ptr())
    ctx.eval_tabular_match(min(n, 4096), min(n, 4096) // 2, 16)
    best = float("inf")
    for _ in range(5):
        t0 = time.perf_counter(); st = ctx.eval_tabular_match(n, (n + 1) // 2, 16); best = min(best, time.perf_counter() - t0)
    m, r2 = int(st[:, 0].sum()), int(st[:, 1].sum())
    by_form["one launch: walks over the deal's tree nodes, statistics summed in the kernel (scopa_eval_tabular_match)"] = {
        "seconds_call": best, "episodes_per_s": n / best, "reward_vs_random": r2 / 2 / m,
        "note": "host clock around the synchronous call (launch + kernel + statistics copy), best of 5; no state traffic per ply: 6 Philox draws and <= 3 threshold compares per episode"}
    return by_form, states


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=1 << 24)
    ap.add_argument("--cfr-iterations", type=int, default=200)
    a = ap.parse_args()
    import torch
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms.vanilla_cfr import CFRTrainer
    from scopa_amd.algorithms.evaluation import evaluate_agent_device
    tr = CFRTrainer(load_game("mini_scopa"))
    tr.train(steps=a.cfr_iterations)
    evaluate_agent_device(tr, 4096)                       # warm-up (kernel load)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reward, stats = evaluate_agent_device(tr, a.episodes)
    wall = time.perf_counter() - t0
    ctx = tr._engine.ctx
    n = a.episodes
    pol = torch.as_tensor(np.ascontiguousarray(ctx.exploitability(return_policy=True)["policy"], np.float64), device="cuda:0")
    by_form, _ = measure_kernels(ctx, pol, n)
    times = by_form["integer thresholds per infoset (scopa_eval_tabular_prepare)"]["seconds_per_ply"]
    k = sum(times)                                     # the form evaluate_agent_device uses (the last one timed)
    print(json.dumps({"kernel": "k_eval_tabular_step", "episodes": n, "policy": f"average policy after {a.cfr_iterations} vanilla-CFR iterations", "reward_vs_random": reward,
                      "reward_std_error": stats["reward_std_error"], "scopas_trained_vs_random": [stats["trained_avg"], stats["opponent_avg"]],
                      "seconds_8_launches": k, "seconds_per_ply": times, "episodes_per_s_kernels": n / k, "episode_plies_per_s": 8 * n / k,
                      "algorithmic_bytes_per_episode_ply": 44, "by_sampling_form": by_form, "achieved_GBps": 8 * n * 44 / k / 1e9, "frac_of_hbm_peak": 8 * n * 44 / k / 1e9 / 8000.0,
                      "evaluate_agent_device_wall_s": wall, "episodes_per_s_evaluate_agent_device": n / wall,
                      "reference_python_episodes_per_s": "~550 (500 episodes every 5 iterations dominate run_mccfr_experiment.py; BASELINE.md section 2)"}))


if __name__ == "__main__":
    main()
