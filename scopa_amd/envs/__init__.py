"""Scopa game environments (reference: src/envs/__init__.py; only MiniScopa is on the solver path)."""
from .mini_scopa_game import Card, MiniDeck, MiniScopaEnv, MiniScopaGame, Player
from . import openspiel_mini_scopa
from .openspiel_mini_scopa import MiniScopaState, PlayerId, load_game

__all__ = ["Card", "MiniDeck", "Player", "MiniScopaGame", "MiniScopaEnv", "MiniScopaState", "PlayerId", "load_game",
           "openspiel_mini_scopa"]
