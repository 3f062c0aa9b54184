#!/usr/bin/env python3
"""The one pin the causal Performer can have: the reference's OWN copy of the causal linear-attention
arithmetic, `StatefulCausalPerformer` in `src/models/perlin_attention/attention_state.py:43-140`.

Run ONLY in the build container (where /root/reference exists):

    python tests/golden/make_golden_performer.py

`performer-pytorch==1.1.4` (FastAttention, the feature map + `causal_linear_attention[_noncuda]`) is a third-party dependency
that is absent from /root/reference and from this image, so step B as a whole stays "parity unpinned".  But the reference
keeps its own restatement of the PREFIX-SUM / DENOMINATOR half:

  * `_causal_linear_attention_noncuda_stateful(q', k', v, eps=1e-6)` (`:97-122`): `D_inv = 1 / (q' . (cumsum(k') + eps))`,
    `out = (cumsum(k' v^T) . q') * D_inv`, fp64 running sums -- the package's `causal_linear_attention_noncuda` with one chunk;
  * `__call__(q', k', v)` (`:53-95`): the same recurrence carried ACROSS calls (kv-cache decoding), chunks of 16, eps 1e-12.

Both take FEATURES (q' = phi(q), k' = phi(k)) -- the reference's stateful path never applies the feature map itself
(`:84-98`).  This script imports that file IN PLACE (stubs only for imports that are absent and unused by the class:
`performer_pytorch`, `numba`-backed `masked_mm`, the transformers-4.32 `hf_bert`), runs both functions on seeded features,
and saves inputs + outputs as `performer.npz`.  Nothing of the reference is copied.

What the fixture pins: the prefix sums, the eps placement and the denominator of this package's `performer.py`
(`causal_linear_attention`) and -- through inputs whose feature map is known in closed form -- of the HIP kernels
(`sea_performer_causal*`).  What stays unpinned: the feature map `phi(x) = relu(d^-1/4 x W^T) + 1e-3` itself
(`generalized_kernel` of the absent package); in the fixture it is computed by this script from the published formula, once
with a Gaussian-orthogonal W and once with a SELECTOR W (rows of the identity: phi reduces to an elementwise expression of q,
so the kernel's feature map is exercised only through a matrix product with 0/1 entries).
"""
import importlib
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load_reference_state_module():
    sys.path.insert(0, REF)

    def stub(name, path=None, **attrs):
        m = types.ModuleType(name)
        if path is not None:
            m.__path__ = [os.path.join(REF, path)]
        m.__package__ = name
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    # parent packages as path-only stubs so that perlin_attention/__init__.py (numba, performer_pytorch, transformers 4.32) never runs
    stub("src", "src"); stub("src.models", "src/models"); stub("src.models.perlin_attention", "src/models/perlin_attention")
    stub("src.models.common", "src/models/common")
    # absent third-party / version-bound modules the class under test never touches
    stub("performer_pytorch", FastAttention=type("FastAttention", (torch.nn.Module,), {}))
    stub("src.models.perlin_attention.masked_mm", sparse_attn=None)
    stub("src.models.hf_bert", BertConfig=type("BertConfig", (), {}))
    mod = importlib.import_module("src.models.perlin_attention.attention_state")
    assert mod.__file__.startswith(REF)
    return mod


def phi(x, W):
    """performer-pytorch 1.1.4 `generalized_kernel(kernel_fn=ReLU, kernel_epsilon=1e-3, normalize_data=True)`: the UNPINNED half."""
    return torch.relu((x.shape[-1] ** -0.25) * x @ W.t()) + 1e-3


def main():
    R = load_reference_state_module()
    sys.path.insert(0, ROOT)
    from sea_attention_amd.perlin_attention.performer import gaussian_orthogonal_random_matrix
    out = {}
    # (name, N, H, T, d, nb, selector W?)
    for name, N, H, T, d, nb, selector in [("selector_d64", 1, 2, 200, 64, 33, True), ("gaussian_d64", 2, 2, 176, 64, 33, False),
                                           ("gaussian_d128", 1, 1, 150, 128, 77, False)]:
        torch.manual_seed(42)
        bf = lambda t: t.to(torch.bfloat16).float()                    # 16-bit representable: one fixture serves fp32 and bf16 kernels
        q, k, v = bf(torch.randn(N, H, T, d) * d ** -0.5), bf(torch.randn(N, H, T, d)), bf(torch.randn(N, H, T, d))
        pos = bf(torch.randn(T, d) * 0.5)                              # v_eye_learned_causal[0, 0, :T]
        if selector:
            W = torch.zeros(nb, d); W[torch.arange(nb), torch.arange(nb)] = 1.0
        else:
            W = bf(gaussian_orthogonal_random_matrix(nb, d))
        qf, kf = phi(q, W), phi(k, W)
        v_aug = torch.cat([pos.view(1, 1, T, d).expand(N, H, T, d), v], -1)   # attention.py:497-514 (value augmentation)
        sp = R.StatefulCausalPerformer(None, None)
        with torch.no_grad():
            full = sp._causal_linear_attention_noncuda_stateful(qf, kf, v_aug)        # eps 1e-6, one chunk
            cut = T - 40
            sp2 = R.StatefulCausalPerformer(None, None)
            pieces = [sp2(qf[..., :cut, :], kf[..., :cut, :], v_aug[..., :cut, :]),
                      sp2(qf[..., cut:cut + 1, :], kf[..., :cut + 1, :], v_aug[..., :cut + 1, :]),
                      sp2(qf[..., cut + 1:, :], kf, v_aug)]                           # eps 1e-12, carried sums (kv-cache form)
        for key, val in dict(q=q, k=k, v=v, pos=pos, W=W, qf=qf, kf=kf, full=full, stateful=torch.cat(pieces, -2)).items():
            out[f"{name}/{key}"] = val.numpy().copy()
        print(name, tuple(full.shape), float(full.abs().max()), float((full - torch.cat(pieces, -2)).abs().max()))
    np.savez_compressed(os.path.join(os.environ.get("GOLDEN_OUT", HERE), "performer.npz"), **out)


if __name__ == "__main__":
    main()
