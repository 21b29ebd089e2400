"""CFR algorithms (reference: src/algorithms/__init__.py -- same exported names)."""
from .vanilla_cfr import CFRTrainer, InfoNode, LearnedCFRPolicy, RandomPolicy
from .mc_cfr import MCCFRTrainer, ScopaLearnedPolicy

__all__ = ["CFRTrainer", "InfoNode", "LearnedCFRPolicy", "RandomPolicy", "MCCFRTrainer", "ScopaLearnedPolicy"]
