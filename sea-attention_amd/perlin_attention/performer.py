"""Causal Performer (FAVOR+, generalized ReLU features) used as SEA's attention estimator.

The reference delegates to a third-party package that is NOT in its tree:
`performer-pytorch==1.1.4` (environment.yml:129), class `FastAttention`, constructed at
src/models/perlin_attention/attention.py:159-164 with causal=True, generalized_attention=True.
PARITY UNPINNED for this step: no reference output exists for it; this file restates the
package's published algorithm --

  phi(x)  = relu(d^-1/4 * x W^T) + 1e-3               (`generalized_kernel`, kernel_epsilon 1e-3)
  out_t   = sum_{s<=t} (phi(q_t).phi(k_s)) v_s / (phi(q_t) . (sum_{s<=t} phi(k_s) + 1e-6))
                                                        (`causal_linear_attention[_noncuda]`, eps 1e-6)
  W       = nb_features x d Gaussian-orthogonal matrix, buffer `projection_matrix`

-- and keeps the buffer name so trained checkpoints (`performer.projection_matrix`) load.

MI355X-first evaluation: the package's fallback materialises the (T, nb, e) outer products and
cumsums them (2.2 GB per batch item at OPT-1.3B); here the prefix is evaluated chunk-wise with
GEMMs (rocBLAS): per chunk an intra-chunk masked (C x C) product plus the running nb x e state.
"""
import math

import torch
from torch import nn


def gaussian_orthogonal_random_matrix(nb_rows, nb_columns, scaling=0):
    blocks = []
    for _ in range(nb_rows // nb_columns):
        q, _ = torch.linalg.qr(torch.randn((nb_columns, nb_columns)), mode='reduced')
        blocks.append(q.t())
    rem = nb_rows - (nb_rows // nb_columns) * nb_columns
    if rem > 0:
        q, _ = torch.linalg.qr(torch.randn((nb_columns, nb_columns)), mode='reduced')
        blocks.append(q.t()[:rem])
    mat = torch.cat(blocks)
    if scaling == 0:
        mult = torch.randn((nb_rows, nb_columns)).norm(dim=1)
    elif scaling == 1:
        mult = math.sqrt(float(nb_columns)) * torch.ones((nb_rows,))
    else:
        raise ValueError(f'Invalid scaling {scaling}')
    return torch.diag(mult) @ mat


def causal_linear_attention(qp, kp, v, chunk=128, eps=1e-6):
    """qp,kp (..., T, nb) >= 0 feature maps, v (..., T, e) -> (..., T, e); fp32 accumulation."""
    *lead, T, nb = qp.shape
    e = v.shape[-1]
    C = min(chunk, T)
    pad = (-T) % C
    if pad:
        qp = torch.nn.functional.pad(qp, (0, 0, 0, pad))
        kp = torch.nn.functional.pad(kp, (0, 0, 0, pad))
        v = torch.nn.functional.pad(v, (0, 0, 0, pad))
    nc = (T + pad) // C
    q_ = qp.reshape(*lead, nc, C, nb)
    k_ = kp.reshape(*lead, nc, C, nb)
    v_ = v.reshape(*lead, nc, C, e)
    kv = torch.matmul(k_.transpose(-1, -2), v_)                   # (..., nc, nb, e) per-chunk state
    state = kv.cumsum(-3) - kv                                    # state BEFORE each chunk
    ks = k_.sum(-2)                                               # (..., nc, nb)
    ksum = ks.cumsum(-2) - ks
    a = torch.matmul(q_, k_.transpose(-1, -2))                    # (..., nc, C, C)
    a = a.tril_()
    num = torch.matmul(a, v_) + torch.matmul(q_, state)
    den = a.sum(-1) + (q_ * ksum.unsqueeze(-2)).sum(-1) + eps * q_.sum(-1)
    out = num / den.unsqueeze(-1)
    return out.reshape(*lead, nc * C, e)[..., :T, :]


class FastAttention(nn.Module):
    def __init__(self, dim_heads, nb_features=None, ortho_scaling=0, causal=False, generalized_attention=False,
                 kernel_fn=None, no_projection=False):
        super().__init__()
        nb_features = nb_features if nb_features is not None else int(dim_heads * math.log(dim_heads))
        self.dim_heads = dim_heads
        self.nb_features = nb_features
        self.ortho_scaling = ortho_scaling
        self.causal = causal
        self.generalized_attention = generalized_attention
        self.kernel_fn = kernel_fn if kernel_fn is not None else nn.ReLU()
        self.no_projection = no_projection
        self.register_buffer('projection_matrix',
                             gaussian_orthogonal_random_matrix(nb_features, dim_heads, ortho_scaling))

    @torch.no_grad()
    def redraw_projection_matrix(self, device):
        self.projection_matrix.copy_(
            gaussian_orthogonal_random_matrix(self.nb_features, self.dim_heads, self.ortho_scaling).to(device))

    def feature_map(self, x):
        d = x.shape[-1]
        proj = self.projection_matrix.to(x.dtype)
        return self.kernel_fn(torch.matmul((d ** -0.25) * x, proj.t())) + 1e-3

    def forward(self, q, k, v):
        if not (self.causal and self.generalized_attention) or self.no_projection:
            raise NotImplementedError("SEA constructs the estimator as causal generalized attention only "
                                      "(attention.py:159-164)")
        out_dtype = v.dtype
        q, k, v = q.float(), k.float(), v.float()
        out = causal_linear_attention(self.feature_map(q), self.feature_map(k), v)
        return out.to(out_dtype)


class ProjectionUpdater(nn.Module):
    """src/models/common/performer.py:5-37: redraw the projection every `feature_redraw_interval`
    training calls.  Present for state-dict compatibility (`calls_since_last_redraw`)."""

    def __init__(self, instance, feature_redraw_interval):
        super().__init__()
        self.instance = instance
        self.feature_redraw_interval = feature_redraw_interval
        self.register_buffer('calls_since_last_redraw', torch.tensor(0))

    def fix_projections_(self):
        self.feature_redraw_interval = None

    def redraw_projections(self, device):
        if not self.training:
            return
        if self.feature_redraw_interval is not None and self.calls_since_last_redraw >= self.feature_redraw_interval:
            for m in self.instance.modules():
                if isinstance(m, FastAttention):
                    m.redraw_projection_matrix(device)
            self.calls_since_last_redraw.zero_()
            return
        self.calls_since_last_redraw += 1

    def forward(self, x):
        raise NotImplementedError
