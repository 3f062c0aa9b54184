"""-m gpu: the row-split SEA layer (SURVEY 8e, N < G: one long sequence over G ranks, estimator included) rehearsed on ONE
GPU -- the G ranks' work run one after another through the same phase functions the multi-GPU runner uses
(distributed.run_row_split_local vs run_row_split) -- against the unsharded layer."""
import pytest
import torch

import sea_attention_amd as S
from sea_attention_amd import distributed as D
from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Cfg:
    def __init__(self, hidden, heads, max_pos):
        self.hidden_size, self.num_attention_heads, self.max_position_embeddings = hidden, heads, max_pos


def _mask_fn(N, dtype):
    fp_min = torch.finfo(torch.float16).min / 2

    def fn(lo, hi):
        rows = torch.arange(lo, hi, device=DEV).view(-1, 1)
        m = ((torch.arange(hi, device=DEV).view(1, hi) > rows) * fp_min).view(1, 1, hi - lo, hi)
        return m.expand(N, 1, hi - lo, hi).contiguous().to(dtype)
    return fn


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("world,T", [(2, 1024), (4, 1024), (4, 832)])
def test_row_split_layer_equals_the_unsharded_layer(dtype, world, T):
    N, H, d, T_M, k = 1, 8, 64, 256, 32
    S.seed(42)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=8, causal=True, k_flatten=True,
                               k_flatten_dim='causal_batch', context_output_method='mix')
    layer = PerlinSelfAttention(Cfg(H * d, H, T), pc).to(DEV).to(dtype).eval()
    for m in layer.modules():
        if hasattr(m, 'benchmarking'):
            m.benchmarking = True
    att = layer.attention
    att.context_layer_dtype = dtype
    # one kernel for steps J-L on both sides: with "auto" the per-block plan may hand a rank whose rows mostly share their
    # keys to the tile kernel as a whole while the unsharded launch stays mixed (same numbers to rounding, not to the bit)
    att.sparse_kernel = "gather"
    S.seed(7)
    x = torch.randn((N, H, T, d), device=DEV).to(dtype)
    q = (x.float() * d ** -0.5).to(dtype)
    mask = _mask_fn(N, dtype)
    with torch.no_grad():
        # the unsharded layer with the Performer cut exactly where the ranks cut (the one-GPU sequence-parallel kernel adds the
        # segments' state increments in order, as the ranks do); world 2: the second rank starts from the first rank's
        # own final state, i.e. from the one-pass kernel's state at that row
        att.performer_segments = 1 if world == 2 else world
        ref = layer(None, None, None, query_layer=q, key_layer=x, value_layer=x, attention_mask=mask(0, T))
        att.performer_segments = 1
        got = D.row_split_layer_local(layer, q, x, x, mask, world)
    att.performer_segments = None
    refc = ref.context_layer.float()
    rel = ((got.float() - refc).norm() / refc.norm()).item()
    same = (got == ref.context_layer).all(-1).float().mean().item()
    print(f"row split world={world} T={T} {dtype}: rel {rel:.2e}, rows bitwise equal {same:.4f}")
    cuts = D.estimator_row_cuts(T, world)
    equal_cuts = len({hi - lo for lo, hi in cuts}) == 1
    if world == 2 or equal_cuts:
        assert torch.equal(got, ref.context_layer), (rel, same)
    else:        # the last rank is shorter than the kernel's equal segments would be: same sums, other grouping
        assert rel < (2e-2 if dtype == torch.bfloat16 else 4e-3) and same > 0.5, (rel, same)
