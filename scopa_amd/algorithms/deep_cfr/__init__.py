from .deep_cfr import AdvantageNetwork, DeepCFR, DeviceMemory, RandomPolicy, StrategyBuffer
from .nets import FlexibleNet, MLPBlock, masked_softmax, positive_regret_policy

__all__ = ["DeepCFR", "AdvantageNetwork", "StrategyBuffer", "DeviceMemory", "RandomPolicy", "FlexibleNet", "MLPBlock",
           "positive_regret_policy", "masked_softmax"]
