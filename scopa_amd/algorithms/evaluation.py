"""Batched on-device evaluation of a tabular policy vs a uniform-random opponent.

The reference's evaluate_agent (vanilla_cfr.py:157-216, mc_cfr.py:146-206) plays episodes one by one in Python and is
where its experiment scripts spend most of their wall time (500 episodes every 5 iterations,
run_mccfr_experiment.py:101-105).  Here all episodes of a match run in one launch as walks over the deal's tree (or, per_ply,
advance in lockstep as packed states stepped by the device step function): the trained seat's action comes from its policy
row (np.random.choice arithmetic) and the opponent's is uniform.  Seats are swapped at half time as in the reference.  Draws are Philox, so the numbers are statistically -- not
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


def _halves_from_sums(st):
    """per seat half of a match, from the integer sums scopa_eval_tabular_match returns (episodes, sum r x2, sum (r x2)^2, scopas of either side)"""
    out = []
    for m, r2, q2, t, o in st.tolist():
        if m:
            var4 = max(m * q2 - r2 * r2, 0) / (m * m)                        # 4 x the variance, from exact integers
            out.append({"episodes": m, "reward": r2 / 2 / m, "reward_std_error": float(np.sqrt(var4) / 2 / np.sqrt(m)),
                        "trained_scopas": t / m, "opponent_scopas": o / m})
        else:
            out.append({"episodes": 0, "reward": 0.0, "reward_std_error": 0.0, "trained_scopas": 0.0, "opponent_scopas": 0.0})
    return out


def evaluate_agent_device(trainer, num_episodes=10000, policy=None, stream_id=16, per_ply=False):
    """-> (avg_reward, scopa_stats) for `trainer`'s average policy (or an explicit [n_infosets][4] table).

    Default: the whole match in ONE launch (scopa_eval_tabular_match) -- every episode plays the trainer's deal, so it is a walk over the deal's
    tree nodes, and the statistics are summed on the device as integers.  per_ply=True: the same episodes bit for bit (same draws, same thresholds)
    as eight launches of the packed-state step kernel with the statistics reduced by torch -- the form whose states stay inspectable between plies."""
    import torch
    eng = trainer._engine
    ctx = eng.ctx
    n = int(num_episodes)
    if policy is None:
        policy = ctx.exploitability(return_policy=True)["policy"]     # the average policy, computed on device
    dev = f"cuda:{ctx.device}"
    pol = torch.as_tensor(np.ascontiguousarray(policy, np.float64), device=dev)
    first = (n + 1) // 2                                              # episodes e < n / 2: the trained agent sits in seat 0 (vanilla_cfr.py:173-176)
    if n == 0:
        return 0.0, {"trained_avg": 0.0, "opponent_avg": 0.0, "difference": 0.0, "data_collected": False, "reward_std_error": 0.0, "by_seat": _halves_from_sums(np.zeros((2, 5), np.int64))}
    torch.cuda.synchronize()
    ctx.eval_tabular_prepare(pol.data_ptr())                  # the policy's sampling thresholds, once: the plies compare integers (same actions, bit for bit)
    if not per_ply:
        st = ctx.eval_tabular_match(n, first, stream_id)
        m, r2, q2, t, o = (int(x) for x in st.sum(axis=0))
        var4 = max(m * q2 - r2 * r2, 0) / (m * m)
        stats = {"trained_avg": t / m, "opponent_avg": o / m, "difference": t / m - o / m, "data_collected": True,
                 "reward_std_error": float(np.sqrt(var4) / 2 / np.sqrt(m)), "by_seat": _halves_from_sums(st)}
        return r2 / 2 / m, stats
    states = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ctx.eval_init_states(states.data_ptr(), n)
    idx = torch.zeros(n, dtype=torch.int32, device=dev)
    seat = (torch.arange(n, device=dev) >= first).to(torch.int32)
    torch.cuda.synchronize()
    for ply in range(8):
        ctx.eval_tabular_step(states.data_ptr(), idx.data_ptr(), n, ply, 0, seat.data_ptr(), stream_id)
    ctx.synchronize()
    # the statistics are reduced on the device (float64; every term is a multiple of 0.5, so the sums are exact): only a dozen numbers
    # cross to the host instead of the 16 n bytes of final states
    b = states.view(torch.uint8).view(n, 16)                                   # scopa_state: ncap at bytes 12, 13; scopas at 14, 15
    r = b[:, 12:14].to(torch.float64) + 2.0 * b[:, 14:16].to(torch.float64)
    total = r.sum(1, keepdim=True)
    rewards = torch.where(total == 0, torch.zeros_like(r), r - total / 2.0)   # evaluate_game (mini_scopa_game.py:106-114)
    sl = seat.long().unsqueeze(1)
    mine = rewards.gather(1, sl).squeeze(1)
    sc = b[:, 14:16].to(torch.float64)
    t_sc, o_sc = sc.gather(1, sl).squeeze(1), sc.gather(1, 1 - sl).squeeze(1)
    by_seat = []
    for lo, hi in ((0, first), (first, n)):
        m = hi - lo
        if m:
            x = mine[lo:hi]
            by_seat.append({"episodes": m, "reward": float(x.mean()), "reward_std_error": float(x.std(unbiased=False) / np.sqrt(m)),
                            "trained_scopas": float(t_sc[lo:hi].mean()), "opponent_scopas": float(o_sc[lo:hi].mean())})
        else:
            by_seat.append({"episodes": 0, "reward": 0.0, "reward_std_error": 0.0, "trained_scopas": 0.0, "opponent_scopas": 0.0})
    stats = {"trained_avg": float(t_sc.mean()), "opponent_avg": float(o_sc.mean()),
             "difference": float(t_sc.mean() - o_sc.mean()), "data_collected": True,
             "reward_std_error": float(mine.std(unbiased=False) / np.sqrt(n)),
             "by_seat": by_seat}
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
