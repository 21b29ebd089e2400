"""OpenSpiel-protocol state / game objects for MiniScopa (mirrors src/envs/openspiel_mini_scopa.py).

`MiniScopaState` offers the pyspiel.State surface the reference's solvers and evaluators use --
current_player, legal_actions, apply_action, is_terminal, is_chance_node, chance_outcomes, rewards, returns,
information_state_string, clone, history_str -- without needing pyspiel; `load_game("mini_scopa")` stands in for
pyspiel.load_game.  Nothing is registered with pyspiel itself (openspiel_mini_scopa.py:166-186 registers a pyspiel.Game
subclass; these are plain Python classes, and pyspiel is not installed where this package is built and tested): callers
switch the one `pyspiel.load_game` line to `scopa_amd.envs.load_game`.
"""
import ctypes as C

from .. import _lib
from .mini_scopa_game import MiniScopaEnv


class PlayerId:
    TERMINAL = -4   # pyspiel.PlayerId.TERMINAL
    CHANCE = -1


class MiniScopaState:
    """State wrapper around MiniScopaEnv (openspiel_mini_scopa.py:5-115)."""

    def __init__(self, game, env=None, num_players=2, skip_reset=False):
        self._game = game
        self.num_players = num_players
        self.env = env or MiniScopaEnv(num_players=num_players)
        if not skip_reset:
            self.env.reset()
        self._is_terminal = False
        self.action_history = []

    def get_game(self):
        return self._game

    def current_player(self):
        if self._is_terminal:
            return PlayerId.TERMINAL
        return self.env.agent_name_mapping[self.env.agent_selection]

    def legal_actions(self, player=None):
        """Action ids of the cards in hand, IN HAND ORDER; [0] if the hand is empty; [] at terminal (:22-47)."""
        if self._is_terminal:
            return []
        if player is None:
            player = self.current_player()
        # from the hand alone, as the reference: the WRAPPER's terminal flag decides above, not the env's (a state built around an env that has already
        # ended, skip_reset=True, still lists the hand -- the flag only follows apply_action, :49-53)
        legal = [c.id for c in self.env.game.players[player].hand]
        return legal if legal else [0]

    def apply_action(self, action):
        self.action_history.append(action)
        self.env.step(action)
        self._is_terminal = all(self.env.terminations.values())

    def _apply_action(self, action):
        self.apply_action(action)

    def is_terminal(self):
        return self._is_terminal

    def is_chance_node(self):
        return False

    def chance_outcomes(self):
        return []

    def history_str(self):
        history_str = "-".join(map(str, self.action_history))
        if self._is_terminal:
            rewards_str = ",".join(f"{r:.2f}" for r in self.rewards())
            return f"TERMINAL:{history_str}:{rewards_str}"
        return f"H:{history_str}:P{self.current_player()}"

    def rewards(self):
        if not self._is_terminal:
            return [0] * self.num_players
        return [self.env.rewards[f"player_{i}"] for i in range(self.num_players)]

    def returns(self):
        return self.rewards()

    def information_state_string(self, player=None):
        """`P{p}:H[hand in order]_T[table in order]` -- list ORDER is part of the key (:86-95)."""
        if player is None:
            player = self.current_player()
        if self._is_terminal or player < 0:
            return "TERMINAL"
        key, buf = C.c_uint64(), C.create_string_buffer(96)         # the key of (player, hand, table) whatever the env's own terminal state: see legal_actions
        _lib.lib().scopa_state_infoset_key(C.byref(self.env.game.packed), int(player), C.byref(key))
        _lib.lib().scopa_key_to_string(key, buf, 96)
        return buf.value.decode()

    def clone(self):
        new_state = MiniScopaState(self._game, env=self.env.clone(), num_players=self.num_players, skip_reset=True)
        new_state._is_terminal = self._is_terminal
        new_state.action_history = self.action_history.copy()
        return new_state

    # -- engine hook: where this state sits in the flat tree -------------------------------------------------
    def tree_path(self):
        """Legal-action INDICES from the root to this state, or None if an illegal (no-op) action was played."""
        s = self._game.new_initial_state() if self.env.seed == getattr(self._game, "seed", 42) else None
        if s is None:
            return None
        path = []
        for a in self.action_history:
            la = s.legal_actions()
            if a not in la:
                return None
            path.append(la.index(a))
            s.apply_action(a)
        return path

    def __str__(self):
        return self.history_str()


class MiniScopaGame:
    """Game object (openspiel_mini_scopa.py:118-159): new_initial_state(), num_players() (a METHOD)."""

    def __init__(self, num_players=2, seed=42):
        self._num_players = num_players
        self.seed = seed                     # the reference hard-wires 42; other deals are the build's extension
        self.perm = _lib.deal_py_seed(seed)
        self.short_name = "mini_scopa"
        self.num_distinct_actions_ = 16
        self.max_game_length_ = num_players * 4

    def num_players(self):
        return self._num_players

    def num_distinct_actions(self):
        return 16

    def max_game_length(self):
        return self._num_players * 4

    def min_utility(self):
        return -10.0

    def max_utility(self):
        return 10.0

    def new_initial_state(self):
        if self.seed == 42:
            return MiniScopaState(self, num_players=self._num_players)
        return MiniScopaState(self, env=MiniScopaEnv(seed=self.seed, num_players=self._num_players),
                              num_players=self._num_players, skip_reset=True)


def _full_scopa(params=None):
    from .openspiel_full_scopa import FullScopaGame
    return FullScopaGame()


def _team_mini_scopa(params=None):
    from .openspiel_team_mini_scopa import TPIMiniScopaGame
    return TPIMiniScopaGame(**(params or {}))


_REGISTRY = {"mini_scopa": lambda params=None: MiniScopaGame(), "full_scopa": _full_scopa, "team_mini_scopa_tpi": _team_mini_scopa}


def load_game(short_name, params=None):
    """Stand-in for pyspiel.load_game for the games this package provides."""
    return _REGISTRY[short_name](params)
