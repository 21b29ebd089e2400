"""Entry script: MCCFR on MiniScopa (mirrors src/mccfr_mini_scopa.py: train 5000 iterations, evaluate vs random).

    python -m scopa_amd.mccfr_mini_scopa              # the reference's sequential semantics, replayed on the GPU
    python -m scopa_amd.mccfr_mini_scopa --batch 4096 # the batched throughput path
"""
import argparse

from scopa_amd.envs import load_game
from scopa_amd.algorithms.mc_cfr import MCCFRTrainer, RandomPolicy, evaluate_agent


def main(iterations=5000, num_episodes=5000, batch=None):
    game = load_game("mini_scopa")
    trainer = MCCFRTrainer(game=game, batch=batch)
    trainer.train(iterations=iterations)
    mc_cfr_policy = trainer.tabular_policy()
    random_policy = RandomPolicy(game)
    avg_reward, avg_reward_history, scopa_stats = evaluate_agent(game, mc_cfr_policy, random_policy, num_episodes=num_episodes)
    print(f"  Info sets learned: {len(trainer.info_sets)}")
    print(f"  Average reward: {avg_reward:.4f}")
    if scopa_stats.get("data_collected", False):
        print("\n  Scopa Statistics:")
        print(f"    Trained agent avg scopas/game:  {scopa_stats['trained_avg']:.4f}")
        print(f"    Random agent avg scopas/game:   {scopa_stats['opponent_avg']:.4f}")
        print(f"    Difference (Trained - Random):  {scopa_stats['difference']:+.4f}")
    print(f"  Exploitability (build-defined): {trainer.exploitability():.6f}")
    return avg_reward


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=5000)
    ap.add_argument("--episodes", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=None)
    a = ap.parse_args()
    main(a.iterations, a.episodes, a.batch)
