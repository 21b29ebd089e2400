#!/usr/bin/env python3
"""Re-run the reference's only published experiment (src/experiments/run_mccfr_experiment.py: 10 MCCFR runs x 500
iterations, evaluation vs random every 5 iterations, final evaluation) on this engine and compare with the band the
reference committed (src/experiments/experiments/results/MiniScopa_MCCFR_data.json; numbers quoted in BASELINE.md §1).
Training uses the reference's sequential semantics (MCCFRTrainer default mode: np.random draws replayed on the GPU);
evaluation is the batched device evaluator.  Output: the reference's tracker `_data.json` schema + a comparison block.

    python benchmarks/reproduce_mccfr_experiment.py --out profiles/r01_MiniScopa_MCCFR_data.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REFERENCE_BAND = {  # BASELINE.md §1 (MiniScopa_MCCFR_data.json:6318-6324 and per-run num_info_sets)
    "final_reward_mean": 1.1545, "final_reward_std": 0.1163, "scopa_trained_mean": 0.4025, "scopa_trained_std": 0.0423,
    "scopa_random_mean": 0.1559, "scopa_random_std": 0.0085, "info_sets_min": 593, "info_sets_max": 732,
    "eval_reward_mean_iter5": 0.4727, "eval_reward_mean_iter500": 1.1293}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--iterations", type=int, default=500)
    ap.add_argument("--eval-interval", type=int, default=5)
    ap.add_argument("--eval-episodes", type=int, default=500)
    ap.add_argument("--final-episodes", type=int, default=5000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from scopa_amd.envs import load_game
    from scopa_amd.algorithms import MCCFRTrainer, evaluate_agent_device
    from scopa_amd.experiment_tracker import ExperimentMetrics, ExperimentTracker
    game = load_game("mini_scopa")
    tracker = ExperimentTracker("MiniScopa_MCCFR")
    t0 = time.perf_counter()
    for run in range(a.runs):
        np.random.seed(1000 + run)
        tr = MCCFRTrainer(game)
        m = ExperimentMetrics(iterations=list(range(a.iterations)), algorithm="MC-CFR")
        for t in range(0, a.iterations, a.eval_interval):
            tr.train(a.eval_interval)
            rew, st = evaluate_agent_device(tr, a.eval_episodes, stream_id=100 + run * 1000 + t)
            m.eval_iterations.append(t + a.eval_interval)
            m.eval_rewards.append(rew)
            m.eval_scopas_trained.append(st["trained_avg"])
            m.eval_scopas_random.append(st["opponent_avg"])
            m.eval_scopa_diff.append(st["difference"])
        rew, st = evaluate_agent_device(tr, a.final_episodes, stream_id=99 + run * 1000)
        m.final_reward, m.final_scopa_trained, m.final_scopa_random = rew, st["trained_avg"], st["opponent_avg"]
        m.final_scopa_diff, m.num_info_sets = st["difference"], len(tr.info_sets)
        tracker.add_run(m)
        print(f"run {run + 1}: final reward {rew:+.4f} scopas {st['trained_avg']:.3f}/{st['opponent_avg']:.3f} infosets {m.num_info_sets}",
              file=sys.stderr)
    data = tracker.plot_data()
    fm = data["statistics"]["final_metrics"]
    info = [r["num_info_sets"] for r in data["runs"]]
    mine = {"final_reward_mean": fm["reward_mean"], "final_reward_std": fm["reward_std"], "scopa_trained_mean": fm["scopa_trained_mean"],
            "scopa_trained_std": fm["scopa_trained_std"], "scopa_random_mean": fm["scopa_random_mean"], "scopa_random_std": fm["scopa_random_std"],
            "info_sets_min": min(info), "info_sets_max": max(info), "eval_reward_mean_iter5": data["statistics"]["rewards"]["mean"][0],
            "eval_reward_mean_iter500": data["statistics"]["rewards"]["mean"][-1]}
    data["comparison_with_reference"] = {"reference": REFERENCE_BAND, "this_engine": mine, "wall_seconds": time.perf_counter() - t0,
                                         "note": "different random seeds on both sides: agreement is statistical (means within ~2 standard errors)"}
    out = json.dumps(data, indent=1)
    if a.out:
        open(a.out, "w").write(out)
    print(json.dumps(data["comparison_with_reference"], indent=1))


if __name__ == "__main__":
    main()
