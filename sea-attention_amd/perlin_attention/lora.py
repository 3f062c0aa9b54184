"""LoRA plumbing used by PerlinSelfAttention (reference: src/models/common/lora.py:6-93).
Training-side feature; kept so the module constructs the same parameters (`*_lora.lora_a/b`)."""
import math

import torch
import torch.nn.functional as F
from torch import nn


class LoraLinear(nn.Module):
    def __init__(self, inch, outch, dim_r):
        super().__init__()
        self.lora_a = nn.Parameter(torch.zeros((dim_r, inch)))
        self.lora_b = nn.Parameter(torch.zeros((outch, dim_r)))
        torch.nn.init.kaiming_uniform_(self.lora_a, a=math.sqrt(5))

    def forward(self, x: torch.Tensor):
        return F.linear(x, torch.mm(self.lora_b, self.lora_a))


def lora_forward_linear(linear: nn.Linear, x: torch.Tensor):
    return F.linear(x, linear.weight, linear.bias)


def _heads_to_hidden(x):
    N, H, T, D = x.shape
    return x.permute(0, 2, 1, 3).reshape(N, T, H * D)


def lora_forward_lora(linear: nn.Linear, linear_x: torch.Tensor, lora: LoraLinear, x: torch.Tensor, enabled: bool):
    """linear_x = linear(x) already computed; add the low-rank update lora(x) under the bias (lora.py:38-93)."""
    if not enabled:
        return linear_x
    op_dtype = linear_x.dtype
    if linear_x.ndim == 4:
        linear_x = _heads_to_hidden(linear_x)
    heads = None
    if x.ndim == 4:
        heads = x.shape
        x = _heads_to_hidden(x)
    bias = linear.bias.view(1, 1, -1) if linear.bias is not None else None
    y = linear_x - bias if bias is not None else linear_x
    y = lora(x) + y.to(op_dtype)
    if bias is not None:
        y = y + bias
    y = y.to(op_dtype)
    if heads is not None:
        N, H, T, D = heads
        y = y.view(N, T, H, D).permute(0, 2, 1, 3).contiguous()
    return y


def lora_forward(linear: nn.Linear, lora: LoraLinear, x: torch.Tensor, enabled: bool):
    """linear(x) with the low-rank update folded in (lora.py:95-99 of the reference)."""
    return lora_forward_lora(linear, lora_forward_linear(linear, x), lora, x, enabled)
