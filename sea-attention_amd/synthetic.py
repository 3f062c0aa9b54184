"""Synthetic compressed attention maps for benchmarks and parity tests (no datasets, no checkpoints here).

Two families, both (N, H, T, T_M) row-stochastic like `estimated_attention_probs` (attention.py:670-673):

* `random_probs`      softmax(randn): every row picks its pixels independently -- the reference's own kernel
                      tests use exactly this (flat_csr_masked_bmm.py:254-267).  Worst case for any reuse of
                      K/V rows between neighbouring queries.
* `structured_probs`  what a trained SEA predictor produces (paper figs. 5/9: a diagonal band of recent keys,
                      a few "vertical" stripes = key positions every query attends to, a sink at key 0, faint
                      noise): neighbouring query rows keep (nearly) the same keys.
"""
import torch


def random_probs(N, H, T, T_M, device, dtype=torch.float32, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.softmax(torch.randn((N, H, T, T_M), device=device, generator=g), -1).to(dtype)


def structured_probs(N, H, T, T_M, device, dtype=torch.float32, seed=0, n_stripes=3, band=2, noise=0.05,
                     T_src=None):
    """Diagonal band + per-head vertical stripes + attention sink, with a little noise on top.

    Row t sees w = T_src - T + t + 1 keys; key kappa falls into pixel floor(kappa * T_M / w), so a vertical stripe is a
    pixel index that drifts with t -- as it does in the reference's interpolation (causal_resize_m_to_t.py:565-569)."""
    T_src = T if T_src is None else T_src
    g = torch.Generator(device=device).manual_seed(seed)
    w = (torch.arange(T, device=device) + (T_src - T + 1)).view(1, 1, T, 1).float()                   # keys visible
    b = torch.arange(T_M, device=device).view(1, 1, 1, T_M).float()
    score = noise * torch.randn((N, H, T, T_M), device=device, generator=g)
    score = score + 6.0 * (b >= T_M - band)                                                          # recent keys
    score = score + 5.0 * (b == 0)                                                                   # sink
    stripes = torch.rand((N, H, n_stripes), device=device, generator=g) * 0.9 * T_src                # key positions
    for i in range(n_stripes):
        kap = stripes[:, :, i].view(N, H, 1, 1)
        pix = torch.floor(kap * T_M / w)                                                             # (N,H,T,1)
        score = score + (4.0 - 0.5 * i) * ((b == pix) & (kap < w))
    return torch.softmax(score, -1).to(dtype)
