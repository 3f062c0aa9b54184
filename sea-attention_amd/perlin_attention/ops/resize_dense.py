"""Dense twin of the mask interpolation (reference: ops/kernels/resize_m_to_t.py:6-73).

Pure torch on whatever device the inputs live on, exactly like the reference (it has no Triton
kernel for this one).  Used by the dense (`benchmarking=False`) mode and as the parity probe for
the CSR path: where the `max_k` clamp is idle both describe the same key set.
"""
import random

import torch
import torch.nn.functional as F


def resize_from_m_to_t(x: torch.Tensor, masked_fill_value: float, attention_mask: torch.Tensor,
                       target_width: int = None, training=False, is_causal=True, k=None, oversampled=None):
    """x (N,H,T1,T_m) -> (N,H,T1,T2): key s of row t reads pixel floor((rank_s + 0.5)/len_t * T_m - 1e-4)
    where rank_s counts the valid keys up to s and len_t the valid keys of the row; invalid keys read
    `masked_fill_value`."""
    assert masked_fill_value is not None
    N, H, T1, T_M = x.shape
    assert attention_mask.shape[1] == 1
    T2 = target_width if target_width is not None else T1

    if is_causal:
        assert attention_mask.ndim == 4
        assert attention_mask.shape == (N, 1, T1, T2)
    else:
        assert attention_mask.ndim == 4
        assert attention_mask.shape == (N, 1, 1, T2), f"{attention_mask.shape} == {T2}"
        attention_mask = attention_mask.expand(N, 1, T1, T2)

    valid = (attention_mask > -1).float()
    rank = valid.cumsum(-1)                       # 1-based rank of each valid key
    length = rank[..., -1:]                       # valid keys per row
    if training and random.random() < 0.1:        # resize_m_to_t.py:39-45: 10 % of calls jitter the ranks
        rank = torch.clamp(rank + (torch.rand_like(rank) * 1.5 - 0.75),
                           torch.ones((1, 1, 1, 1), device=x.device),
                           rank.max(dim=-1, keepdim=True)[1])
    pixel = torch.floor((rank - 0.5) / length * T_M - 1e-4).to(torch.long)
    pixel = pixel + ((1 - valid) * T_M).to(torch.long)      # invalid keys -> the fill column
    pixel = pixel.clamp(0, T_M).expand(N, H, T1, T2)
    out = F.pad(x, (0, 1), value=masked_fill_value).gather(-1, pixel)

    if oversampled is not None:
        # resize_m_to_t.py:54-71: an oversampled compressed mask is thinned again at full width
        assert isinstance(oversampled, (float, int))
        assert isinstance(k, (int, float))
        xs = torch.arange(0, T2, device=length.device).view(1, 1, 1, T2)
        ps = torch.clamp_min(torch.round(length / oversampled), 1)
        oys = torch.clamp(length, round(k), round(k * oversampled)) / k
        pos = (xs + 1) / length * ps
        keep = torch.abs(pos - torch.round(pos)) <= ((1 / oys) * 0.5 + 1e-4)
        out.masked_fill_(~keep, value=masked_fill_value)
    return out
