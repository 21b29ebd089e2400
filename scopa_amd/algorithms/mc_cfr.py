"""External-sampling MCCFR with the reference's interface (mirrors src/algorithms/mc_cfr.py).

Two modes behind one class:
  * default (`batch=None`): the reference's own sequential semantics.  `iteration()` draws the 463 uniforms the
    reference's np.random.choice calls would draw from the GLOBAL numpy stream and replays them on the device
    (k_mccfr_replay): tables come out bit-identical to the reference under the same np.random.seed.
  * `batch=B`: the throughput path (k_mccfr_traverse): B traversals per traverser per iteration against tables
    frozen for the iteration, Philox draws.  A documented semantic difference, validated against the
    oracle's definition and by exploitability, not by equality with the reference's tables.
"""
import numpy as np

from ..engine import Engine
from .vanilla_cfr import _Policy

DRAWS_PER_ITERATION = 463  # decision visits per iteration(): one np.random.choice each (mc_cfr.py:55)


class InfoNode:
    def __init__(self, legal_actions, regret_sum=None, strategy_sum=None):
        self.legal_actions = np.asarray(legal_actions)
        n = self.legal_actions.size
        self.regret_sum = np.zeros(n) if regret_sum is None else regret_sum
        self.strategy_sum = np.zeros(n) if strategy_sum is None else strategy_sum

    def current_strategy(self):
        pos = np.maximum(self.regret_sum, 0)
        if pos.sum() == 0:
            return np.ones_like(pos) / len(pos)
        return pos / pos.sum()


class MCCFRTrainer:
    def __init__(self, game, batch=None, seed=0x5C09A, device=0):
        self.game = game
        self.batch = batch
        self._engine = Engine(game, device=device)
        self._engine.ctx.mccfr_seed(seed)
        self._map = {}
        self._stale = False

    def iteration(self):
        """One pass per player (mc_cfr.py:88-92)."""
        self._run(1)

    def train(self, iterations=10000):
        self._run(iterations)
        return []

    def _run(self, iterations):
        ctx = self._engine.ctx
        if self.batch is None:
            chunk = 2000  # uniforms are drawn from the global numpy stream exactly as the reference consumes them
            done = 0
            while done < iterations:
                k = min(chunk, iterations - done)
                u = np.random.random_sample(DRAWS_PER_ITERATION * k)
                used = ctx.mccfr_replay(k, u)
                assert used == u.size
                done += k
        else:
            ctx.mccfr_iterate(self.batch, iterations)
        self._stale = True

    @property
    def info_sets(self):
        """dict[(player, infoset string) -> InfoNode], only infosets visited so far, in first-visit order."""
        if self._stale or not self._map:
            e = self._engine
            R, S, _ = e.ctx.tables_get(local=False)
            self._map = {}
            for i in e.visited_order():
                n = int(e.nlegal[i])
                self._map[(int(e.player[i]), e.keys[i])] = InfoNode(e.legal[i, :n].copy(), R[i, :n].copy(), S[i, :n].copy())
            self._stale = False
        return self._map

    def tabular_policy(self):
        return ScopaLearnedPolicy(self.game, self.info_sets)

    def exploitability(self):
        from .exploitability import exploitability_of_tables
        return exploitability_of_tables(self._engine)


class ScopaLearnedPolicy(_Policy):
    def __init__(self, game, info_sets):
        super().__init__(game, list(range(game.num_players())))
        self.info_sets = info_sets

    def action_probabilities(self, state):
        if state.is_terminal():
            return {}
        player = state.current_player()
        key = (player, state.information_state_string(player))
        if key in self.info_sets:
            node = self.info_sets[key]
            total = node.strategy_sum.sum()
            if total > 1e-12:
                probs = node.strategy_sum / total
            else:
                probs = np.ones(len(node.legal_actions)) / len(node.legal_actions)
            return {action: probs[i] for i, action in enumerate(node.legal_actions)}
        legal = state.legal_actions(player)
        prob = 1.0 / len(legal)
        return {action: prob for action in legal}


class RandomPolicy(_Policy):
    def __init__(self, game):
        super().__init__(game, list(range(game.num_players())))

    def action_probabilities(self, state):
        if state.is_terminal():
            return {}
        player = state.current_player()
        legal_actions = state.legal_actions(player)
        prob = 1.0 / len(legal_actions)
        return {action: prob for action in legal_actions}


def evaluate_agent(game, trained_policy, opponent_policy, num_episodes=10000):
    """mc_cfr.py:146-206 (identical to the vanilla evaluator plus the two-player check)."""
    if game.num_players() != 2:
        raise ValueError("evaluate_agent only supports 2-player games")
    from .evaluation import head_to_head
    return head_to_head(game, trained_policy, opponent_policy, num_episodes)
