"""Team MiniScopa in its Team-Public-Information form (two coordinators, four seats) as an OpenSpiel-protocol state over
the packed 40-byte engine.  Mirrors src/envs/openspiel_team_mini_scopa.py (TPIMiniScopaState, TPIMiniScopaGame) and the
parts of src/envs/team_mini_scopa_game.py its callers touch (`state.env.game.players[i].{hand,captures,scopas,team_id}`,
`.table`, `.last_capture_team`, `env.agent_selection`, `env.rewards`).  No reference solver uses this game; it is the
state engine only.  Card repr follows the team module (`rank_suit`, team_mini_scopa_game.py:12-13)."""
from .. import _lib
from .mini_scopa_game import RANKS, SUITS, card_id


class Card:
    def __init__(self, rank, suit):
        self.rank, self.suit = rank, suit

    @classmethod
    def from_id(cls, cid):
        suit = SUITS[cid >> 2]
        return cls(RANKS[suit][cid & 3], suit)

    @property
    def id(self):
        return card_id(self.rank, self.suit)

    def __repr__(self):
        return f"{self.rank}_{self.suit}"

    def __eq__(self, other):
        return self.rank == other.rank and self.suit == other.suit

    def __hash__(self):
        return hash((self.rank, self.suit))


class _PlayerView:
    def __init__(self, name, team_id):
        self.name, self.team_id, self.hand, self.captures, self.scopas = name, team_id, [], [], 0


class _GameView:
    def __init__(self, snap):
        self.players = [_PlayerView(f"player_{i}", i // 2) for i in range(4)]
        for i, p in enumerate(self.players):
            p.hand = [Card.from_id(c) for c in snap["hands"][i]]
            p.captures = [Card.from_id(c) for c in snap["caps"][i]]
            p.scopas = snap["scopas"][i]
        self.table = [Card.from_id(c) for c in snap["table"]]
        self.last_capture_team = None if snap["last"] < 0 else snap["last"]

    def get_team(self, player_id):
        return self.players[player_id].team_id


class _EnvView:
    possible_agents = [f"player_{i}" for i in range(4)]
    agent_name_mapping = {f"player_{i}": i for i in range(4)}
    max_steps = 16

    def __init__(self, state, seed):
        self._state, self.seed = state, seed
        self.agents = self.possible_agents[:]

    @property
    def game(self):
        return _GameView(_lib.unpack_team_state(self._state._ts.s[0]))

    @property
    def step_count(self):
        return int(self._state._ts.s[0]["step"])

    @property
    def agent_selection(self):
        return f"player_{self._state._ts.seat()}"

    @property
    def rewards(self):
        r = self._state._ts.player_rewards() if self._state._ts.is_terminal() else [0, 0, 0, 0]
        return {f"player_{i}": r[i] for i in range(4)}

    @property
    def terminations(self):
        t = self._state._ts.is_terminal()
        return {a: t for a in self.possible_agents}


class TPIMiniScopaState:
    def __init__(self, game, seed=42):
        self._game = game
        self._ts = _lib.TeamState(seed=seed)
        self._dead = []      # actions applied AFTER the end of the game: dead steps for the env (team_mini_scopa_game.py:174-176), but the reference's
                             # action_history -- and with it history_str and the A[...] part of every infoset string -- still grows (openspiel_team_mini_scopa.py:88-92)
        self.env = _EnvView(self, seed)

    def get_game(self):
        return self._game

    @property
    def action_history(self):
        return self._ts.history() + self._dead

    def current_player(self):
        """the coordinator (team) of the seat to move; PlayerId.TERMINAL (-4) at the end"""
        return self._ts.current_player()

    def legal_actions(self, player=None):
        return self._ts.legal()

    def apply_action(self, action):
        if self._ts.is_terminal():
            self._dead.append(int(action))
            return
        self._ts.step(action)

    def is_terminal(self):
        return self._ts.is_terminal()

    def is_chance_node(self):
        return False

    def chance_outcomes(self):
        return []

    def rewards(self):
        return self._ts.rewards()

    def returns(self):
        return self.rewards()

    def information_state_string(self, player):
        s = self._ts.infoset_string(player)
        if self._dead and s.endswith("]"):
            tail = "-".join(map(str, self._dead))
            s = s[:-1] + ("-" if not s.endswith("A[]") else "") + tail + "]"
        return s

    def history_str(self):
        if not self._dead:
            return self._ts.history_str()
        h = "-".join(map(str, self.action_history))
        return f"TERMINAL:{h}:" + ",".join(f"{r:.2f}" for r in self.rewards())

    def clone(self):
        c = TPIMiniScopaState.__new__(TPIMiniScopaState)
        c._game, c._ts, c._dead = self._game, self._ts.copy(), list(self._dead)
        c.env = _EnvView(c, self.env.seed)
        return c


class TPIMiniScopaGame:
    def __init__(self, seed=42):
        self.seed = seed

    def num_players(self):
        return 2  # two teams

    def num_distinct_actions(self):
        return 16

    def max_game_length(self):
        return 16

    def new_initial_state(self):
        return TPIMiniScopaState(self, seed=self.seed)
