"""MiniScopa game objects with the reference's public interface, backed by the packed 16-byte state.

Mirrors src/envs/mini_scopa_game.py of rug-marl-group2/scopa (Card, MiniDeck, Player, MiniScopaGame,
MiniScopaEnv): same attribute names, same call signatures, same quirks (every env is the seed-42 deal unless a
seed is given, mini_scopa_game.py:25-28,131-132; an action whose card is not in hand is a silent no-op that
still consumes the turn, :155-159).  All rule evaluation goes through libscopa_hip.so's scopa_state_* entry points
(the same __host__ __device__ rules the kernels run); this module only keeps the object views callers touch.
No pettingzoo / gymnasium dependency.
"""
import ctypes as C

import numpy as np

from .. import _lib

SUITS = ["cuori", "fiori", "picche", "bello"]
RANKS = {"cuori": [2, 5, 8, 10], "fiori": [2, 5, 7, 9], "picche": [3, 6, 8, 9], "bello": [3, 6, 7, 10]}


def card_id(rank, suit):
    """Card id = action id = suit_idx*4 + rank_idx (mini_scopa_game.py:149-153)."""
    return SUITS.index(suit) * 4 + RANKS[suit].index(rank)


class Card:
    def __init__(self, rank: int, suit: str):
        self.rank = rank
        self.suit = suit

    @classmethod
    def from_id(cls, cid):
        suit = SUITS[cid >> 2]
        return cls(RANKS[suit][cid & 3], suit)

    @property
    def id(self):
        return card_id(self.rank, self.suit)

    def __repr__(self):
        return f"{self.rank}_of_{self.suit}"

    def __eq__(self, other):
        return isinstance(other, Card) and (self.rank, self.suit) == (other.rank, other.suit)

    def __hash__(self):
        return hash((self.rank, self.suit))


class MiniDeck:
    """16-card deck; MiniDeck(seed).cards is the CPython random.seed(seed)+shuffle order (mini_scopa_game.py:25-28),
    reproduced by scopa_deal_py_seed without touching the global `random` state."""
    suits = SUITS
    ranks = RANKS

    def __init__(self, seed=42):
        self.perm = _lib.deal_py_seed(seed)
        self.cards = [Card.from_id(int(c)) for c in self.perm]

    def deal(self, n):
        dealt = self.cards[:n]
        self.cards = self.cards[n:]
        return dealt


class Player:
    def __init__(self, name):
        self.name = name
        self.hand = []
        self.captures = []
        self.scopas = 0

    def reset(self):
        self.hand.clear()
        self.captures.clear()
        self.scopas = 0


class _Discrete:
    def __init__(self, n):
        self.n = n


class MiniScopaGame:
    """Holds the packed state (`.packed`, a scopa_state) and keeps the reference's object views in sync."""

    def __init__(self, num_players=2):
        if num_players != 2:
            raise ValueError("the packed MiniScopa engine is two-player (as every reference entry point is)")
        self.num_players = num_players
        self.players = [Player(f"player_{i}") for i in range(num_players)]
        self.table = []
        self.last_capture = None
        self.packed = _lib.State16()
        self.perm = None
        self.reset(42)

    def reset(self, seed=42):
        self.deck = MiniDeck(seed)
        self.perm = self.deck.perm.copy()
        rc = _lib.lib().scopa_state_init(self.perm.ctypes.data_as(C.c_void_p), C.byref(self.packed))
        if rc:
            raise _lib.ScopaError(rc, "scopa_state_init")
        for p in self.players:
            p.reset()
            p.hand = self.deck.deal(4)
        self.table = []
        self.last_capture = None

    # -- views <- packed -----------------------------------------------------------------------
    def _sync_from_packed(self):
        s = self.packed
        for i, p in enumerate(self.players):
            p.hand = [Card.from_id((s.hand[i] >> (4 * k)) & 15) for k in range(s.nh[i])]
            p.scopas = int(s.scopas[i])
        self.table = [Card.from_id((s.table >> (4 * k)) & 15) for k in range(s.nt)]

    def play_action(self, action, player_idx):
        """One MiniScopaEnv.step worth of rules on the packed state; returns True if a card was played."""
        s = self.packed
        before_table = [(s.table >> (4 * k)) & 15 for k in range(s.nt)]
        had = any(((s.hand[player_idx] >> (4 * k)) & 15) == action for k in range(s.nh[player_idx]))
        rc = _lib.lib().scopa_state_step(C.byref(s), int(action))
        if rc:
            raise _lib.ScopaError(rc, "scopa_state_step")
        if had:
            after = [(s.table >> (4 * k)) & 15 for k in range(s.nt)]
            player = self.players[player_idx]
            if len(after) <= len(before_table) and action not in after:  # a capture happened (play_card :95-101)
                captured = [c for c in before_table if c not in after]
                player.captures.extend([Card.from_id(c) for c in captured] + [Card.from_id(action)])
                self.last_capture = player
        self._sync_from_packed()
        return had

    def card_in_table(self, card):
        """(isin, captured cards) for playing `card` on the current table (mini_scopa_game.py:66-91)."""
        probe = _lib.State16.from_buffer_copy(self.packed)
        probe.step = 0
        probe.hand[0] = card.id
        probe.nh[0] = 1
        probe.nh[1] = max(probe.nh[1], 1)
        before = [(probe.table >> (4 * k)) & 15 for k in range(probe.nt)]
        _lib.lib().scopa_state_step(C.byref(probe), card.id)
        after = [(probe.table >> (4 * k)) & 15 for k in range(probe.nt)]
        if card.id in after and len(after) == len(before) + 1:
            return False, []
        return True, [Card.from_id(c) for c in before if c not in after]

    def evaluate_game(self):
        """+1 per captured card, +2 per scopa, centred to zero-sum (mini_scopa_game.py:106-114)."""
        rewards = [len(p.captures) + 2 * p.scopas for p in self.players]
        total = sum(rewards)
        if total == 0:
            return [0] * self.num_players
        mean = total / self.num_players
        return [r - mean for r in rewards]


class MiniScopaEnv:
    """AEC-style environment (reset / step / get_state / set_state) -- mini_scopa_game.py:117-194."""
    metadata = {"name": "Mini-Scopa-v0"}

    def __init__(self, seed=42, num_players=2):
        self.num_players = num_players
        self.game = MiniScopaGame(num_players=num_players)
        self.possible_agents = [f"player_{i}" for i in range(num_players)]
        self.agent_name_mapping = {name: i for i, name in enumerate(self.possible_agents)}
        self._action_spaces = {a: _Discrete(16) for a in self.possible_agents}
        self.max_steps = num_players * 4
        self.seed = seed
        self.reset(seed)

    def action_space(self, agent):
        return self._action_spaces[agent]

    def reset(self, seed=None):
        self.game.reset(seed or self.seed)  # `seed or self.seed`: a falsy seed means "the env's own" (:132)
        self.agents = self.possible_agents[:]
        self.agent_selection = self.agents[0]
        self.rewards = {a: 0 for a in self.agents}
        self.terminations = {a: False for a in self.agents}
        self.truncations = {a: False for a in self.agents}
        self.step_count = 0

    def step(self, action):
        if self.terminations[self.agent_selection]:
            return  # _was_dead_step
        agent = self.agent_selection
        idx = self.agent_name_mapping[agent]
        self._sync_limit()
        self.game.play_action(int(action), idx)
        self.step_count = int(self.game.packed.step) & _lib.STEP_COUNT_MASK
        if _lib.lib().scopa_state_is_terminal(C.byref(self.game.packed)):
            r = self.game.evaluate_game()
            for i, a in enumerate(self.agents):
                self.rewards[a] = r[i]
                self.terminations[a] = True
        self.agent_selection = self.agents[(self.agents.index(agent) + 1) % self.num_players]

    def _sync_limit(self):
        """env.max_steps -> the packed state: 8 for a fresh env (mini_scopa_game.py:127), 16 for a clone's (openspiel_mini_scopa.py:108);
        the terminal rule (:160) reads it from the state's SCOPA_STEP_CLONED bit."""
        if self.max_steps not in (8, 16):
            raise ValueError("MiniScopaEnv.max_steps is 8 (fresh env) or 16 (clone) in the reference; the packed state carries no other limit")
        s = self.game.packed
        s.step = (s.step & _lib.STEP_COUNT_MASK) | (_lib.STEP_CLONED if self.max_steps == 16 else 0)

    def get_state(self):
        g = self.game
        return {
            "table": [(c.rank, c.suit) for c in g.table],
            "hands": [[(c.rank, c.suit) for c in p.hand] for p in g.players],
            "captures": [[(c.rank, c.suit) for c in p.captures] for p in g.players],
            "scopas": [p.scopas for p in g.players],
            "agent_selection": self.agent_selection,
            "step_count": self.step_count,
            "agents": self.agents[:],
            "rewards": dict(self.rewards),
            "terminations": dict(self.terminations),
            "truncations": dict(self.truncations),
        }

    def set_state(self, state):
        g = self.game
        g.table = [Card(r, s) for r, s in state["table"]]
        s = g.packed
        s.table = sum(c.id << (4 * k) for k, c in enumerate(g.table))
        s.nt = len(g.table)
        for i, p in enumerate(g.players):
            p.hand = [Card(r, su) for r, su in state["hands"][i]]
            p.captures = [Card(r, su) for r, su in state["captures"][i]]
            p.scopas = state["scopas"][i]
            s.hand[i] = sum(c.id << (4 * k) for k, c in enumerate(p.hand))
            s.nh[i] = len(p.hand)
            s.ncap[i] = len(p.captures)
            s.scopas[i] = p.scopas
        self.agent_selection = state["agent_selection"]
        self.step_count = state["step_count"]
        s.step = self.step_count
        self._sync_limit()                 # set_state leaves max_steps as it is (mini_scopa_game.py:184-194)
        self.agents = state["agents"][:]
        self.rewards = dict(state["rewards"])
        self.terminations = dict(state["terminations"])
        self.truncations = dict(state["truncations"])

    def clone(self):
        """Cheap copy used by MiniScopaState.clone(): 16 bytes + the capture lists.  As in the reference (openspiel_mini_scopa.py:97-115)
        the copy's max_steps is 16 (:108), not num_players * 4: a clone fed illegal no-op actions plays on past step 8."""
        e = MiniScopaEnv.__new__(MiniScopaEnv)
        e.num_players = self.num_players
        e.possible_agents = self.possible_agents
        e.agent_name_mapping = self.agent_name_mapping
        e._action_spaces = self._action_spaces
        e.max_steps = 16
        e.seed = self.seed
        g = MiniScopaGame.__new__(MiniScopaGame)
        g.num_players = self.game.num_players
        g.players = [Player(p.name) for p in self.game.players]
        for q, p in zip(g.players, self.game.players):
            q.captures = list(p.captures)
        g.packed = _lib.State16()
        rc = _lib.lib().scopa_state_clone(C.byref(self.game.packed), C.byref(g.packed))
        if rc:
            raise _lib.ScopaError(rc, "scopa_state_clone")
        g.perm = self.game.perm
        g.deck = self.game.deck
        g.last_capture = None
        g.table = []
        g._sync_from_packed()
        e.game = g
        e.agents = self.agents[:]
        e.agent_selection = self.agent_selection
        e.rewards = dict(self.rewards)
        e.terminations = dict(self.terminations)
        e.truncations = dict(self.truncations)
        e.step_count = self.step_count
        return e
