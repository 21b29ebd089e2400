"""Batched on-device evaluation of a tabular policy vs a uniform-random opponent.

The reference's evaluate_agent (vanilla_cfr.py:157-216, mc_cfr.py:146-206) plays episodes one by one in Python and is
where its experiment scripts spend most of their wall time (500 episodes every 5 iterations,
run_mccfr_experiment.py:101-105).  Here all episodes advance in lockstep: the packed states are stepped by the device
step function, the trained seat's action comes from its policy row (np.random.choice arithmetic) and the opponent's is
uniform.  Seats are swapped at half time as in the reference.  Draws are Philox, so the numbers are statistically -- not
bitwise -- equivalent to the reference's np.random stream; `evaluate_agent` in vanilla_cfr / mc_cfr remains the
bit-reproducing host version."""
import numpy as np

from .. import _lib


# ---- host evaluator shared by vanilla_cfr.evaluate_agent and mc_cfr.evaluate_agent ---------------------------------------
class MatchTally:
    """Running means of a head-to-head match from the trained agent's point of view: reward, scopas of either side.  `result`
    gives the reference evaluators' return shape (vanilla_cfr.py:157-216): (avg_reward, reward history, scopa_stats)."""

    def __init__(self):
        self.sums = np.zeros(3)          # reward, trained scopas, opponent scopas
        self.curves = ([], [], [], [])   # running means of: reward, trained scopas, opponent scopas, scopa difference

    def add(self, reward, own_scopas, opp_scopas):
        self.sums += (reward, own_scopas, opp_scopas)
        n = len(self.curves[0]) + 1
        w, t, o = (x / n for x in self.sums.tolist())
        for curve, x in zip(self.curves, (w, t, o, (self.sums[1] - self.sums[2]) / n)):
            curve.append(x)

    def result(self, num_episodes):
        w, t, o = (x / num_episodes for x in self.sums.tolist())
        rewards, trained, opponent, diff = self.curves
        return w, rewards, {"trained_avg": t, "opponent_avg": o, "difference": t - o,
                            "history": {"trained": trained, "opponent": opponent, "diff": diff},
                            "data_collected": len(trained) > 0}


def play_out(state, seat_policies, choose):
    """Advance `state` to the end: at every ply the seat's policy gives {action: probability} and `choose(actions, p=probs)`
    picks (np.random.choice in the reference-compatible evaluators: ONE draw per ply, so a seeded match reproduces the reference's)."""
    while not state.is_terminal():
        dist = seat_policies[state.current_player()].action_probabilities(state)
        state.apply_action(choose(tuple(dist.keys()), p=tuple(dist.values())))
    return state


def head_to_head(game, trained_policy, opponent_policy, num_episodes, choose=None):
    """`num_episodes` games of trained vs opponent, the trained agent in seat 0 for the first half and in seat 1 afterwards."""
    choose = np.random.choice if choose is None else choose
    tally = MatchTally()
    for episode in range(num_episodes):
        seat = 0 if episode < num_episodes / 2 else 1
        lineup = (trained_policy, opponent_policy) if seat == 0 else (opponent_policy, trained_policy)
        end = play_out(game.new_initial_state(), lineup, choose)
        scopas = [p.scopas for p in end.env.game.players]
        tally.add(end.rewards()[seat], scopas[seat], scopas[1 - seat])
    return tally.result(num_episodes)


def evaluate_agent_device(trainer, num_episodes=10000, policy=None, stream_id=16):
    """-> (avg_reward, scopa_stats) for `trainer`'s average policy (or an explicit [n_infosets][4] table)."""
    import torch
    eng = trainer._engine
    ctx = eng.ctx
    n = int(num_episodes)
    if policy is None:
        policy = ctx.exploitability(return_policy=True)["policy"]     # the average policy, computed on device
    dev = f"cuda:{ctx.device}"
    pol = torch.as_tensor(np.ascontiguousarray(policy, np.float64), device=dev)
    states = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ctx.eval_init_states(states.data_ptr(), n)
    idx = torch.zeros(n, dtype=torch.int32, device=dev)
    seat_h = np.array([0 if e < n / 2 else 1 for e in range(n)], np.int32)
    seat = torch.as_tensor(seat_h, device=dev)
    torch.cuda.synchronize()
    for ply in range(8):
        ctx.eval_tabular_step(states.data_ptr(), idx.data_ptr(), n, ply, pol.data_ptr(), seat.data_ptr(), stream_id)
    ctx.synchronize()
    raw = states.cpu().numpy().view(_lib.STATE_DTYPE).reshape(-1)
    r = raw["ncap"].astype(np.int64) + 2 * raw["scopas"].astype(np.int64)
    total = r.sum(1)
    rewards = np.where(total[:, None] == 0, 0.0, r - total[:, None] / 2.0)   # evaluate_game (mini_scopa_game.py:106-114)
    ar = np.arange(n)
    mine = rewards[ar, seat_h]
    t_sc = raw["scopas"][ar, seat_h].astype(np.float64)
    o_sc = raw["scopas"][ar, 1 - seat_h].astype(np.float64)
    stats = {"trained_avg": float(t_sc.mean()), "opponent_avg": float(o_sc.mean()),
             "difference": float(t_sc.mean() - o_sc.mean()), "data_collected": n > 0,
             "reward_std_error": float(mine.std() / np.sqrt(max(n, 1))),
             "by_seat": match_halves(mine, t_sc, o_sc, seat_h)}
    return float(mine.mean()), stats


def match_halves(reward, own_scopas, opp_scopas, seat):
    """The two halves of a seat-swapped match apart (the trained agent in seat 0 / in seat 1): episodes, mean reward and its standard
    error, mean scopas of either side -- what an exact tree enumeration of "policy vs uniform from that seat" can be held against."""
    out = []
    for s in (0, 1):
        k = seat == s
        m = int(k.sum())
        out.append({"episodes": m, "reward": float(reward[k].mean()) if m else 0.0,
                    "reward_std_error": float(reward[k].std() / np.sqrt(m)) if m else 0.0,
                    "trained_scopas": float(own_scopas[k].mean()) if m else 0.0, "opponent_scopas": float(opp_scopas[k].mean()) if m else 0.0})
    return out
