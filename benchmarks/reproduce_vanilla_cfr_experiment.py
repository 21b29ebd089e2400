#!/usr/bin/env python3
"""The reference's vanilla-CFR experiment protocol (src/experiments/run_vanilla_cfr_experiment.py:59-131: per iteration one
`_cfr_recursive` traversal per player driven directly, a 500-episode evaluation vs random every `eval_interval` iterations, a final
evaluation, everything into the tracker's `_data.json` schema) on this engine.  Training runs on the GPU (scheduled exact CFR,
bit-identical tables); the evaluations are the host evaluators, which consume the global numpy stream as the reference does -- so
`np.random.seed(k)` reproduces the reference's numbers exactly (tests/golden/vanilla_cfr_experiment.json).

    python benchmarks/reproduce_vanilla_cfr_experiment.py --iterations 500 --out profiles/MiniScopa_VanillaCFR_data.json
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_vanilla_cfr_experiment(game, iterations=500, eval_interval=5, final_eval_episodes=5000, quick_episodes=500):
    """-> ExperimentMetrics with the fields the reference's runner fills."""
    from scopa_amd.algorithms.vanilla_cfr import CFRTrainer, RandomPolicy, evaluate_agent
    from scopa_amd.algorithms.evaluation import head_to_head
    from scopa_amd.experiment_tracker import ExperimentMetrics
    trainer, random_policy = CFRTrainer(game=game), RandomPolicy(game)
    m = ExperimentMetrics(iterations=list(range(iterations)), algorithm="Vanilla CFR")
    for t in range(iterations):
        for player in range(game.num_players()):
            trainer._cfr_recursive(game.new_initial_state(), player, 1.0, 1.0)
        if (t + 1) % eval_interval == 0:
            reward, _, st = head_to_head(game, trainer.get_openspiel_policy(), random_policy, quick_episodes)
            m.eval_iterations.append(t + 1)
            m.eval_rewards.append(reward)
            m.eval_scopas_trained.append(st["trained_avg"])
            m.eval_scopas_random.append(st["opponent_avg"])
            m.eval_scopa_diff.append(st["trained_avg"] - st["opponent_avg"])
    m.final_reward, _, st = evaluate_agent(game, trainer.get_openspiel_policy(), random_policy, num_episodes=final_eval_episodes)
    m.final_scopa_trained, m.final_scopa_random, m.final_scopa_diff = st["trained_avg"], st["opponent_avg"], st["difference"]
    m.num_info_sets = len(trainer.info_set_map)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=500)
    ap.add_argument("--eval-interval", type=int, default=5)
    ap.add_argument("--final-episodes", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from scopa_amd.envs import load_game
    from scopa_amd.experiment_tracker import ExperimentTracker
    if a.seed is not None:
        np.random.seed(a.seed)
    t0 = time.perf_counter()
    m = run_vanilla_cfr_experiment(load_game("mini_scopa"), a.iterations, a.eval_interval, a.final_episodes)
    tracker = ExperimentTracker("MiniScopa_VanillaCFR")
    tracker.add_run(m)
    if a.out:
        tracker.save_data_for_plotting(a.out)
    print(f"final reward {m.final_reward:+.4f}  scopas {m.final_scopa_trained:.3f}/{m.final_scopa_random:.3f}  infosets {m.num_info_sets}  "
          f"({time.perf_counter() - t0:.1f} s)")


if __name__ == "__main__":
    main()
