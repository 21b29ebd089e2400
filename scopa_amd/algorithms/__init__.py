"""CFR algorithms (reference: src/algorithms/__init__.py -- same exported names)."""
from .vanilla_cfr import CFRTrainer, InfoNode, LearnedCFRPolicy, RandomPolicy
from .mc_cfr import MCCFRTrainer, ScopaLearnedPolicy
from .evaluation import evaluate_agent_device

__all__ = ["CFRTrainer", "InfoNode", "LearnedCFRPolicy", "RandomPolicy", "MCCFRTrainer", "ScopaLearnedPolicy", "evaluate_agent_device"]
