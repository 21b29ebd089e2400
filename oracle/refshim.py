"""Container-only stand-ins for the third-party *base classes* the reference imports.

TEST INFRASTRUCTURE — never imported by the product (`scopa_amd/`), by `bench.py`'s
timed path or on the GPU box.  Used only by `oracle/gen_golden.py`, which runs the
reference's own Python from /root/reference to produce the committed fixtures under
`tests/golden/`.

`pyspiel`, `open_spiel`, `pettingzoo` and `gymnasium` are not installed in this image
(no network).  On the MiniScopa path they contribute only base classes and a game
registry -- no arithmetic -- so empty stand-ins let the reference's own
game / CFR / MCCFR / Deep-CFR code run unmodified.  The one third-party *computation*
on the path, `open_spiel.python.algorithms.exploitability.exploitability`
(/root/reference/src/algorithms/vanilla_cfr.py:115), is NOT stood in for: calling it
raises, and exploitability therefore stays "parity unpinned" (DESIGN.md).
"""
import sys
import types


def install():
    if "pyspiel" in sys.modules:
        return
    sys.dont_write_bytecode = True

    # --- gymnasium.spaces.Discrete -------------------------------------------------
    gymnasium = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Discrete:
        def __init__(self, n):
            self.n = n

    spaces.Discrete = Discrete
    gymnasium.spaces = spaces

    # --- pettingzoo.AECEnv ---------------------------------------------------------
    pettingzoo = types.ModuleType("pettingzoo")

    class AECEnv:
        def __init__(self):
            pass

        def _was_dead_step(self, action):
            return None

    pettingzoo.AECEnv = AECEnv

    # --- pyspiel -------------------------------------------------------------------
    pyspiel = types.ModuleType("pyspiel")

    class _NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class GameType(_NS):
        class Dynamics:
            SEQUENTIAL = "sequential"
            SIMULTANEOUS = "simultaneous"

        class ChanceMode:
            DETERMINISTIC = "deterministic"
            EXPLICIT_STOCHASTIC = "explicit"
            SAMPLED_STOCHASTIC = "sampled"

        class Information:
            IMPERFECT_INFORMATION = "imperfect"
            PERFECT_INFORMATION = "perfect"
            ONE_SHOT = "one_shot"

        class Utility:
            ZERO_SUM = "zero_sum"
            CONSTANT_SUM = "constant_sum"
            GENERAL_SUM = "general_sum"
            IDENTICAL = "identical"

        class RewardModel:
            TERMINAL = "terminal"
            REWARDS = "rewards"

    class GameInfo(_NS):
        pass

    class PlayerId:
        TERMINAL = -4
        CHANCE = -1
        SIMULTANEOUS = -2
        INVALID = -3

    class State:
        def __init__(self, game):
            self._game = game

        def get_game(self):
            return self._game

    class Game:
        def __init__(self, game_type, game_info, params):
            self._type, self._info, self._params = game_type, game_info, params

        def get_type(self):
            return self._type

    _registry = {}

    def register_game(game_type, factory):
        _registry[game_type.short_name] = factory

    def load_game(name, params=None):
        return _registry[name](params)

    pyspiel.GameType, pyspiel.GameInfo, pyspiel.PlayerId = GameType, GameInfo, PlayerId
    pyspiel.State, pyspiel.Game = State, Game
    pyspiel.register_game, pyspiel.load_game = register_game, load_game

    # --- open_spiel.python.policy / .algorithms.exploitability ----------------------
    open_spiel = types.ModuleType("open_spiel")
    os_python = types.ModuleType("open_spiel.python")
    os_policy = types.ModuleType("open_spiel.python.policy")
    os_algs = types.ModuleType("open_spiel.python.algorithms")
    os_expl = types.ModuleType("open_spiel.python.algorithms.exploitability")

    class Policy:
        def __init__(self, game, player_ids):
            self.game, self.player_ids = game, player_ids

    def exploitability(game, policy):
        raise NotImplementedError("open_spiel is absent: exploitability is parity-unpinned")

    os_policy.Policy = Policy
    os_expl.exploitability = exploitability
    os_python.policy, os_python.algorithms = os_policy, os_algs
    os_algs.exploitability = os_expl
    open_spiel.python = os_python

    # --- tqdm (installed, but silence the bars) is left alone -----------------------
    for name, mod in {
        "gymnasium": gymnasium, "gymnasium.spaces": spaces, "pettingzoo": pettingzoo,
        "pyspiel": pyspiel, "open_spiel": open_spiel, "open_spiel.python": os_python,
        "open_spiel.python.policy": os_policy, "open_spiel.python.algorithms": os_algs,
        "open_spiel.python.algorithms.exploitability": os_expl,
    }.items():
        sys.modules[name] = mod


def import_reference(root="/root/reference"):
    """Import the reference's hot-path modules. Returns a namespace of modules."""
    import os
    os.environ.setdefault("MPLBACKEND", "Agg")
    os.environ.setdefault("TQDM_DISABLE", "1")
    install()
    for p in (root + "/src", root, root + "/src/algorithms/deep_cfr"):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib
    ns = types.SimpleNamespace()
    ns.game = importlib.import_module("envs.mini_scopa_game")
    ns.spiel = importlib.import_module("envs.openspiel_mini_scopa")
    ns.vanilla = importlib.import_module("algorithms.vanilla_cfr")
    ns.mc = importlib.import_module("algorithms.mc_cfr")
    return ns
