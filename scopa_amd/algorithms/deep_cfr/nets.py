"""Advantage-network building blocks (mirrors the part of src/algorithms/deep_cfr/nets.py that any caller uses:
FlexibleNet(mode="mlp"), MLPBlock, positive_regret_policy, masked_softmax -- nets.py:80-101,151-235,296-331).
State-dict keys match the reference (backbone.<i>.fc.{weight,bias}, head.{weight,bias}), so checkpoints of either
load into the other.  The conv2d_mlp mode is not on the MiniScopa path (no caller) and is not provided."""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

_ACTS = {"relu": nn.ReLU, "gelu": nn.GELU, "tanh": nn.Tanh, "elu": nn.ELU, "leaky_relu": nn.LeakyReLU, "silu": nn.SiLU,
         "identity": nn.Identity, "none": nn.Identity}


def _norm_1d(kind, dim):
    if kind in ("none", None):
        return nn.Identity()
    if kind in ("layer", "layernorm"):
        return nn.LayerNorm(dim)
    if kind in ("batch", "batchnorm"):
        return nn.BatchNorm1d(dim)
    raise ValueError(f"unknown norm {kind}")


def masked_softmax(logits: torch.Tensor, mask: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    very_neg = torch.tensor(-1e9, dtype=logits.dtype, device=logits.device)
    masked = torch.where(mask > 0, logits, very_neg)
    probs = torch.softmax(masked, dim=-1)
    z = (probs * mask).sum(dim=-1, keepdim=True).clamp_min(eps)
    return (probs * mask) / z


def positive_regret_policy(adv: torch.Tensor, mask: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """Regret matching; an all-zero row (not uniform) when no advantage is positive (nets.py:93-101)."""
    pos = torch.relu(adv) * mask
    z = pos.sum(dim=-1, keepdim=True).clamp_min(eps)
    return pos / z


class MLPBlock(nn.Module):
    def __init__(self, in_dim, out_dim, act="relu", norm="none", dropout=0.0, residual=False):
        super().__init__()
        self.fc = nn.Linear(in_dim, out_dim)
        self.norm = _norm_1d(norm, out_dim)
        self.act = _ACTS[act]()
        self.drop = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        self.residual = residual and (in_dim == out_dim)

    def forward(self, x):
        y = self.drop(self.act(self.norm(self.fc(x))))
        return y + x if self.residual else y


class FlexibleNet(nn.Module):
    def __init__(self, input_shape: Tuple[int, ...], output_dim: int, mode: str = "mlp", mlp_hidden: Optional[List[int]] = None,
                 mlp_act: str = "relu", mlp_norm: str = "none", mlp_dropout: float = 0.0, mlp_residual: bool = False, **conv_kwargs):
        super().__init__()
        if mode != "mlp":
            raise NotImplementedError("only mode='mlp' is on the MiniScopa solver path (the reference's conv2d_mlp has no caller)")
        assert len(input_shape) == 1, "For 'mlp', input_shape must be (D,)."
        self.mode = mode
        layers, last = [], input_shape[0]
        for h in (mlp_hidden or []):
            layers.append(MLPBlock(last, h, act=mlp_act, norm=mlp_norm, dropout=mlp_dropout, residual=mlp_residual))
            last = h
        self.backbone = nn.Sequential(*layers)
        self.head = nn.Linear(last, output_dim)

    def forward(self, x):
        return self.head(self.backbone(x))
