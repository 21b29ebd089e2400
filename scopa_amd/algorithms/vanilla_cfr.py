"""Vanilla CFR with the reference's interface and its exact numbers (mirrors src/algorithms/vanilla_cfr.py).

CFRTrainer.train / ._cfr_recursive run on the GPU (k_cfr_exact: the reference's sequential DFS semantics,
bit-identical tables); InfoNode / LearnedCFRPolicy / RandomPolicy / evaluate_agent are host-side objects with
the reference's attributes and the reference's use of np.random, so seeded evaluations reproduce too.
"""
import numpy as np

from ..engine import Engine


class InfoNode:
    """Per-infoset arrays (vanilla_cfr.py:8-39).  Arrays are host snapshots of the device tables."""

    def __init__(self, legal_actions, regret_sum=None, strategy_sum=None, local_strategy=None):
        self.legal_actions = np.asarray(legal_actions)
        n = self.legal_actions.size
        self.regret_sum = np.zeros(n) if regret_sum is None else regret_sum
        self.strategy_sum = np.zeros(n) if strategy_sum is None else strategy_sum
        self.local_strategy = np.ones(n) / n if local_strategy is None else local_strategy

    def get_strategy(self):
        positive_regrets = np.maximum(self.regret_sum, 0)
        norm_sum = np.sum(positive_regrets)
        if norm_sum > 0:
            return positive_regrets / norm_sum
        return np.ones(self.legal_actions.size) / self.legal_actions.size

    @property
    def policy(self) -> np.ndarray:
        norm_sum = np.sum(self.strategy_sum)
        if norm_sum > 0:
            return self.strategy_sum / norm_sum
        return np.ones(self.legal_actions.size) / self.legal_actions.size


class CFRTrainer:
    """`CFRTrainer(game).train(steps)`; `.info_set_map`: dict[infoset string -> InfoNode] in first-visit order."""

    def __init__(self, game, device=0, mode="exact"):
        """mode="exact": the reference's sequential semantics (bit-identical tables).  mode="sync": textbook
        simultaneous-update CFR (strategy frozen per iteration, level-parallel kernel) -- same fixed point, not the
        reference's trajectory."""
        if mode not in ("exact", "sync"):
            raise ValueError("mode must be 'exact' or 'sync'")
        self.game = game
        self.mode = mode
        self._engine = Engine(game, device=device)
        self._map = {}
        self._stale = False

    # -- reference surface ---------------------------------------------------------------------------------
    def _cfr_recursive(self, state, traversing_player, reach_p0, reach_p1):
        """One traversal from `state` (vanilla_cfr.py:56-99) on the device; returns the node value."""
        if state.is_terminal():
            return state.rewards()[traversing_player]
        path = state.tree_path() if hasattr(state, "tree_path") else None
        if path is None:
            raise ValueError("_cfr_recursive: the state is not a node of this game's tree (illegal action played or other deal)")
        self._stale = True
        return self._engine.ctx.cfr_exact_traverse_from(traversing_player, path, reach_p0, reach_p1)

    def train(self, steps: int, eval_interval: int = 1000, compute_exploitability: bool = False):
        """vanilla_cfr.py:105-120.  Exploitability here is the build's own device implementation (the reference
        calls OpenSpiel's, which is absent: parity unpinned) and is returned as [(iteration, value), ...]."""
        exploitability_history = []
        ctx = self._engine.ctx
        done = 0
        while done < steps:
            chunk = steps - done
            if compute_exploitability:
                chunk = min(chunk, eval_interval - (done % eval_interval))
            if self.mode == "exact":
                ctx.cfr_exact_iterate(chunk)
            else:
                ctx.cfr_sync_iterate(chunk)
            done += chunk
            self._stale = True
            if compute_exploitability and done % eval_interval == 0:
                exploitability_history.append((done, self.exploitability()))
        return exploitability_history

    def exploitability(self):
        from .exploitability import exploitability_of_tables
        return exploitability_of_tables(self._engine)

    @property
    def info_set_map(self):
        if self._stale or not self._map:
            self._refresh()
        return self._map

    def get_openspiel_policy(self):
        return LearnedCFRPolicy(self.game, self.info_set_map)

    # -- device -> host views --------------------------------------------------------------------------------
    def _refresh(self):
        e = self._engine
        R, S, L = e.ctx.tables_get()
        self._map = {}
        for i in e.visited_order():
            n = int(e.nlegal[i])
            self._map[e.keys[i]] = InfoNode(e.legal[i, :n].copy(), R[i, :n].copy(), S[i, :n].copy(), L[i, :n].copy())
        self._stale = False


class _Policy:
    def __init__(self, game, player_ids):
        self.game = game
        self.player_ids = player_ids


class LearnedCFRPolicy(_Policy):
    def __init__(self, game, info_set_map):
        super().__init__(game, list(range(game.num_players())))
        self.info_set_map = info_set_map

    def action_probabilities(self, state):
        if state.is_terminal():
            return {}
        player = state.current_player()
        info_state = state.information_state_string(player)
        legal_actions = state.legal_actions()
        if info_state in self.info_set_map:
            probs = self.info_set_map[info_state].policy
            return {action: probs[i] for i, action in enumerate(legal_actions)}
        prob = 1.0 / len(legal_actions)
        return {action: prob for action in legal_actions}


class RandomPolicy(_Policy):
    """Uniform over legal actions (vanilla_cfr.py:146-155)."""

    def __init__(self, game):
        super().__init__(game, list(range(game.num_players())))

    def action_probabilities(self, state):
        legal_actions = state.legal_actions()
        prob = 1.0 / len(legal_actions)
        return {action: prob for action in legal_actions}


def evaluate_agent(game, trained_policy, opponent_policy, num_episodes=10000):
    """-> (avg_reward, running-mean reward history, scopa_stats), the return shape of vanilla_cfr.py:157-216; draws come from the
    global np.random stream exactly as there (one np.random.choice per ply), so a seeded run reproduces the reference's numbers."""
    from .evaluation import head_to_head
    return head_to_head(game, trained_policy, opponent_policy, num_episodes)
