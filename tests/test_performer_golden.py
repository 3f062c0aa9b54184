"""The causal Performer's prefix-sum / denominator half against the REFERENCE's own code: tests/golden/performer.npz holds
what `StatefulCausalPerformer._causal_linear_attention_noncuda_stateful` and `StatefulCausalPerformer.__call__`
(src/models/perlin_attention/attention_state.py:43-122, imported in place by tests/golden/make_golden_performer.py) return for
seeded features phi(q), phi(k) and the augmented values.

CPU: `perlin_attention.performer.causal_linear_attention` (this package's chunked evaluation) on the same features.
GPU: `sea_performer_causal*` through `ops.performer_value` / `ops.performer_step` on the q, k, v, W the features were made
from.  The feature map itself (`generalized_kernel` of performer-pytorch 1.1.4, absent) is the half no reference code pins; the
`selector` case reduces it to an elementwise expression (W = rows of the identity)."""
import os

import numpy as np
import pytest
import torch

from sea_attention_amd.perlin_attention import performer as PF

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "performer.npz"))
CASES = ["selector_d64", "gaussian_d64", "gaussian_d128"]


def _t(case, key):
    return torch.from_numpy(G[f"{case}/{key}"])


@pytest.mark.parametrize("case", CASES)
def test_prefix_sum_stage_equals_reference_function(case):
    qf, kf, v, pos, full = (_t(case, n) for n in ("qf", "kf", "v", "pos", "full"))
    N, H, T, d = v.shape
    v_aug = torch.cat([pos.view(1, 1, T, d).expand(N, H, T, d), v], -1)
    for chunk in (128, 16, T):
        got = PF.causal_linear_attention(qf, kf, v_aug, chunk=chunk)
        torch.testing.assert_close(got, full, atol=2e-5, rtol=2e-5)
    # the kv-cache form (eps 1e-12, sums carried across three calls) describes the same rows
    torch.testing.assert_close(_t(case, "stateful"), full, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("case", CASES)
def test_module_with_the_fixtures_projection(case):
    """`FastAttention.forward` = published feature map + the pinned prefix sums."""
    q, k, v, pos, W, full = (_t(case, n) for n in ("q", "k", "v", "pos", "W", "full"))
    N, H, T, d = v.shape
    fa = PF.FastAttention(d, W.shape[0], causal=True, generalized_attention=True)
    fa.projection_matrix.copy_(W)
    torch.testing.assert_close(fa.feature_map(q), _t(case, "qf"), atol=1e-6, rtol=1e-6)
    out = fa(q, k, torch.cat([pos.view(1, 1, T, d).expand(N, H, T, d), v], -1))
    torch.testing.assert_close(out, full, atol=2e-5, rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_hip_performer_equals_reference_function(case, dtype):
    from sea_attention_amd.perlin_attention import ops
    q, k, v, pos, W, full = (_t(case, n).cuda() for n in ("q", "k", "v", "pos", "W", "full"))
    N, H, T, d = v.shape
    if not ops.performer_supported(d, W.shape[0]):
        pytest.skip("shape outside the fused Performer kernel")
    if dtype == torch.float16:
        q, k, v, pos = (t.clamp(-60000, 60000) for t in (q, k, v, pos))
    out = ops.performer_value(q.to(dtype), k.to(dtype), v.to(dtype), pos.to(dtype), W)
    assert out.shape == (N, H, T, 3 * d)
    got = out[..., :2 * d].float()
    # inputs are bf16-representable: the 16-bit kernels see exactly the fixture's q, k, v (fp16 rounds them once more); what is
    # left is their split-operand MFMA arithmetic and the 16-bit rounding of the stored result
    tol = {torch.float32: 2e-4, torch.bfloat16: 2e-2, torch.float16: 4e-3}[dtype]
    err = (got - full).abs().max().item()
    assert err <= tol * max(1.0, full.abs().max().item()), err
    assert torch.equal(out[..., 2 * d:].float(), v.to(dtype).float())
    for nseg in (2,):                                                 # the sequence-parallel form (two segments fit every T here): same rows
        seg = ops.performer_value(q.to(dtype), k.to(dtype), v.to(dtype), pos.to(dtype), W, n_segments=nseg)
        assert (seg[..., :2 * d].float() - full).abs().max().item() <= tol * max(1.0, full.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["gaussian_d64", "gaussian_d128"])
def test_hip_performer_step_equals_reference_stateful_call(case):
    """kv-cache form: `sea_performer_causal_step` in the fixture's three pieces (prefix, one row, the rest) against
    `StatefulCausalPerformer.__call__`'s rows for the same pieces."""
    from sea_attention_amd.perlin_attention import ops
    dtype = torch.bfloat16
    q, k, v, pos, W, ref = (_t(case, n).cuda() for n in ("q", "k", "v", "pos", "W", "stateful"))
    N, H, T, d = v.shape
    if not ops.performer_avg_supported(q.to(dtype), W.shape[0]) or not ops.performer_chunk_rows(d, W.shape[0], dtype):
        pytest.skip("no stateful step for this shape")
    q, k, v, pos = (t.to(dtype) for t in (q, k, v, pos))
    cut, state, rows = T - 40, None, []
    for t0, t1 in ((0, cut), (cut, cut + 1), (cut + 1, T)):
        o, _, state = ops.performer_step(q[:, :, t0:t1], k[:, :, :t1], v[:, :, :t1], pos, W, state_in=state, t_base=t0)
        rows.append(o[..., :2 * d].float())
    got = torch.cat(rows, -2)
    assert (got - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
