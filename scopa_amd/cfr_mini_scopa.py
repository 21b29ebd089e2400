"""Entry script: vanilla CFR on MiniScopa (mirrors src/cfr_mini_scopa.py: train 500 steps, evaluate vs random)."""
from scopa_amd.envs import load_game
from scopa_amd.algorithms.vanilla_cfr import CFRTrainer, RandomPolicy, evaluate_agent


def plot(avg_reward_history, scopa_stats=None, path="cfr_miniscopa_final_performance.png"):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return
    plt.figure(figsize=(15, 5))
    plt.subplot(1, 2, 1)
    plt.plot(avg_reward_history, linewidth=1.5)
    plt.axhline(y=0, color="r", linestyle="--", alpha=0.5, label="Break-even")
    plt.title("Average Reward vs. Random Agent"); plt.xlabel("Games Played"); plt.ylabel("Average Reward")
    plt.legend(); plt.grid(True)
    plt.subplot(1, 2, 2)
    hist = scopa_stats["history"]
    plt.plot(hist["trained"], label="Trained Agent", linewidth=1.5, alpha=0.8)
    plt.plot(hist["opponent"], label="Random Agent", linewidth=1.5, alpha=0.8)
    plt.title("Average Scopas per Game"); plt.xlabel("Games Played"); plt.ylabel("Average Scopas")
    plt.legend(); plt.grid(True)
    plt.tight_layout()
    plt.savefig(path, dpi=150)


def main(steps=500, num_episodes=500, do_plot=True):
    game = load_game("mini_scopa")
    trainer = CFRTrainer(game=game)
    trainer.train(steps=steps, eval_interval=5, compute_exploitability=False)
    cfr_policy = trainer.get_openspiel_policy()
    random_policy = RandomPolicy(game)
    avg_reward, avg_reward_history, scopa_stats = evaluate_agent(game, cfr_policy, random_policy, num_episodes=num_episodes)
    if do_plot:
        plot(avg_reward_history, scopa_stats)
    print(f"  Info sets learned: {len(trainer.info_set_map)}")
    print(f"  Average reward: {avg_reward:.4f}")
    print(f"  Trained agent avg scopas/game:  {scopa_stats['trained_avg']:.4f}")
    print(f"  Random agent avg scopas/game:   {scopa_stats['opponent_avg']:.4f}")
    print(f"  Exploitability (build-defined): {trainer.exploitability():.6f}")
    return avg_reward


if __name__ == "__main__":
    main()
