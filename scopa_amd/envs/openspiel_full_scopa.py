"""FullScopa (40-card) OpenSpiel-protocol state over the packed engine (mirrors src/envs/openspiel_full_scopa.py and the
parts of src/envs/full_scopa_game.py its callers touch).  No reference solver uses this game; it is the state engine
only."""
import numpy as np

from .. import _lib

SUITS = ["denari", "coppe", "spade", "bastoni"]


class Card:
    def __init__(self, rank, suit):
        self.rank, self.suit = rank, suit

    @classmethod
    def from_id(cls, cid):
        return cls(cid % 10 + 1, SUITS[cid // 10])

    @property
    def id(self):
        return SUITS.index(self.suit) * 10 + self.rank - 1

    def __repr__(self):
        return f"{self.rank}_{self.suit}"

    def __eq__(self, other):
        return self.rank == other.rank and self.suit == other.suit

    def __hash__(self):
        return hash((self.rank, self.suit))


class _PlayerView:
    def __init__(self, name):
        self.name, self.hand, self.captures, self.scopas = name, [], [], 0


class _GameView:
    """`.players[i].{hand,captures,scopas}`, `.table`, `.round_number` as the reference's evaluators read them."""

    def __init__(self, snap):
        self.players = [_PlayerView(f"player_{i}") for i in range(2)]
        for i, p in enumerate(self.players):
            p.hand = [Card.from_id(c) for c in snap["hands"][i]]
            p.captures = [Card.from_id(c) for c in snap["caps"][i]]
            p.scopas = snap["scopas"][i]
        self.table = [Card.from_id(c) for c in snap["table"]]
        self.round_number = snap["round"]


class _EnvView:
    def __init__(self, state):
        self._state = state

    @property
    def game(self):
        return _GameView(self._state._fs.snapshot())

    @property
    def step_count(self):
        return int(self._state._fs.s[0]["step"])


class FullScopaState:
    def __init__(self, game, seed=42):
        self._game = game
        self.num_players = 2
        self._fs = _lib.FullState(seed=seed)
        self.action_history = []
        self.env = _EnvView(self)

    def get_game(self):
        return self._game

    def current_player(self):
        return self._fs.current_player()

    def legal_actions(self, player=None):
        return self._fs.legal(-1 if player is None else player)

    def apply_action(self, action):
        self.action_history.append(action)
        self._fs.step(action)

    def is_terminal(self):
        return self._fs.is_terminal()

    def is_chance_node(self):
        return False

    def chance_outcomes(self):
        return []

    def rewards(self):
        return self._fs.rewards()

    def returns(self):
        return self.rewards()

    def information_state_string(self, player):
        return self._fs.infoset_string(player)

    def history_str(self):
        h = "-".join(map(str, self.action_history))
        if self.is_terminal():
            return f"TERMINAL:{h}:" + ",".join(f"{r:.2f}" for r in self.rewards())
        return f"H:{h}:P{self.current_player()}"

    def clone(self):
        c = FullScopaState.__new__(FullScopaState)
        c._game, c.num_players = self._game, 2
        c._fs = _lib.FullState.__new__(_lib.FullState)
        c._fs.deck = self._fs.deck
        c._fs.s = self._fs.s.copy()
        c.action_history = self.action_history.copy()
        c.env = _EnvView(c)
        return c


class FullScopaGame:
    def __init__(self, num_players=2, seed=42):
        if num_players != 2:
            raise ValueError("FullScopa is provided for two players")
        self._num_players, self.seed = num_players, seed

    def num_players(self):
        return self._num_players

    def num_distinct_actions(self):
        return 40

    def new_initial_state(self):
        return FullScopaState(self, seed=self.seed)
