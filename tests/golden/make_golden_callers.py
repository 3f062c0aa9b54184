#!/usr/bin/env python3
"""Golden vectors from the reference's OWN CALLERS running on THIS package.

Run ONLY in the build container (where /root/reference exists):

    python tests/golden/make_golden_callers.py

north_star: "... exposed through the same perlin attention-module/operator API so src.trainer.perlin_trainer and
src/models drop onto it unchanged".  This script does that drop: it aliases `src.models.perlin_attention{,.modules,.ops}`
to `sea_attention_amd.perlin_attention` (the alias INTEGRATION.md section 1 shows), imports the reference's
`src/models/perlin_opt/perlin_opt.py` IN PLACE and UNCHANGED, builds its `OPTAttention` / `OPTDecoderLayer`
(`perlin_opt.py:137-239,638-800`) -- which construct `PerlinSelfAttention` from this package -- and runs them on the CPU in
dense mode (`benchmarking=False`; the sparse mode has no CPU path by design).  Inputs, a recipe for the state dict and the
reference caller's outputs are saved as `callers.npz`; `tests/test_reference_callers.py` replays the same inputs through
`opt_plumbing.SeaOPTAttention` / `SeaOPTDecoderLayer` (CPU: exact; `-m gpu`: sparse mode on the MI355X, stateless and decoded piecewise through
`past_key_value = (k, v, state)`, against the same outputs).

Not in the fixture: a CPU run WITH `past_key_value`.  Without `use_cache` the reference asserts `T_DST == T_SRC`
(`PA/attention.py:404-406`) and this package does the same; with `use_cache=True` this package's cached forward is the sparse
mode's (kernels, MI355X only: there is deliberately no CPU path), so the reference's caller cannot be driven through it here.
The cached path is instead checked ON THE GPU against this fixture's stateless output (`y_attn`).

Nothing of the reference is copied: the file holds arrays only.  Harness stubs, all for third-party imports that are absent
from this image and OFF the SEA path (`attention_method='perlin'` never touches them):
  * `turtle`                                   (`perlin_opt.py:19` imports `hideturtle`; needs tkinter)
  * `sinkhorn_transformer.sinkhorn_transformer` (`perlin_opt.py:185`)
  * `reformer_pytorch.reformer_pytorch`         (`perlin_opt.py:213`)
The layer surgery of the reference's benchmark (`src/main/benchmark_bert.py:162-203`: `attention_method`, `benchmarking` on
every module that has it, `fc1` / `fc2` / `out_proj` -> `nn.Identity`) is applied to the reference's layer object here,
attribute by attribute, as `exam` does (`exam` itself needs a GPU for its timing loop and `AutoConfig.from_pretrained`
needs the hub).
"""
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
from torch import nn


def install_alias_and_stubs():
    import sea_attention_amd                                           # noqa: F401
    from sea_attention_amd import perlin_attention
    sys.modules["src.models.perlin_attention"] = perlin_attention
    sys.modules["src.models.perlin_attention.modules"] = perlin_attention.modules
    sys.modules["src.models.perlin_attention.ops"] = perlin_attention.ops

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Absent:
        def __init__(self, *a, **kw):
            raise RuntimeError("off-path third-party module, stubbed by tests/golden/make_golden_callers.py")
    stub("turtle", hideturtle=lambda *a, **kw: None)
    stub("sinkhorn_transformer").sinkhorn_transformer = stub("sinkhorn_transformer.sinkhorn_transformer", SinkhornCausalAttention=_Absent)
    stub("reformer_pytorch").reformer_pytorch = stub("reformer_pytorch.reformer_pytorch", LSHAttention=_Absent)
    sys.path.insert(0, REF)


def state_dict_recipe(keys_shapes, seed):
    """The SAME function lives in tests/test_reference_callers.py: a state dict from a seed, key by key in sorted order, so the
    fixture carries (names, shapes, seed) instead of tens of MB of weights.  1-D tensors named like a norm's weight stay
    near 1, everything else ~ N(0, 0.05) (dense-mode softmaxes stay well conditioned)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in sorted(keys_shapes):
        t = torch.randn(shape, generator=g) * 0.05
        if k.endswith("weight") and len(shape) == 1:
            t = t + 1.0
        sd[k] = t
    return sd


def exam_mask(N, T_dst, T_src):
    """benchmark_bert.py:190-194: (1 - tril) * -32000, expanded over the batch."""
    t = torch.arange(T_dst).view(-1, 1) + (T_src - T_dst)
    s = torch.arange(T_src).view(1, -1)
    return ((s > t) * -32000.0).view(1, 1, T_dst, T_src).expand(N, 1, T_dst, T_src).contiguous()


# (name, hidden, heads, T, N, k, predictor_length)
CASES = [
    ("tiny",    64,  2,  48, 2, 4, 16),
    ("opt125m", 768, 12, 128, 1, 64, 256),      # OPT-125m's attention shape (H = 12, d = 64), BASELINE cfg 1's k / w / nbf
]


def main():
    install_alias_and_stubs()
    from sea_attention_amd.perlin_attention import PerlinAttentionConfig, register_default_config
    import importlib
    ref_opt = importlib.import_module("src.models.perlin_opt.perlin_opt")
    from transformers.models.opt.configuration_opt import OPTConfig
    assert ref_opt.__file__.startswith(REF), ref_opt.__file__
    out = {}
    for name, hidden, heads, T, N, k, w in CASES:
        tag = name
        register_default_config(PerlinAttentionConfig(
            k=k, attention_predictor_length=w, performer_nb_factor=8, k_flatten=True, k_flatten_dim="causal_batch",
            causal=True, context_output_method="mix"))
        cfg = OPTConfig(hidden_size=hidden, num_attention_heads=heads, ffn_dim=hidden, max_position_embeddings=T,
                        num_hidden_layers=1, do_layer_norm_before=True, dropout=0.0, attention_dropout=0.0,
                        word_embed_proj_dim=hidden)
        torch.manual_seed(0)
        layer = ref_opt.OPTDecoderLayer(cfg).eval()              # the reference's class, unchanged
        attn = layer.self_attn
        assert type(attn).__module__ == "src.models.perlin_opt.perlin_opt"
        assert type(attn.perlin_self_attention).__module__.startswith("sea_attention_amd."), type(attn.perlin_self_attention)
        keys_shapes = [(k_, tuple(v_.shape)) for k_, v_ in layer.state_dict().items()]
        seed = 1234
        sd = state_dict_recipe(keys_shapes, seed)
        missing, unexpected = layer.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
        for m in layer.modules():                                  # benchmark_bert.py:162-173 (dense mode: benchmarking False)
            if isinstance(m, ref_opt.OPTAttention):
                m.attention_method = "perlin"
        g = torch.Generator().manual_seed(7)
        x = torch.randn((N, T, hidden), generator=g)
        mask = exam_mask(N, T, T)
        with torch.no_grad():
            y_attn, _, present = attn(hidden_states=layer.self_attn_layer_norm(x), attention_mask=mask)
            y_layer = layer(hidden_states=x, attention_mask=mask)[0]
            # the benchmark's surgery (benchmark_bert.py:196-203), dense mode so that the CPU can run it
            fc1 = nn.Identity(); fc1.weight = layer.fc1.weight; layer.fc1 = fc1
            layer.fc2 = nn.Identity()
            op = nn.Identity(); op.weight = attn.out_proj.weight; attn.out_proj = op
            y_exam = layer(hidden_states=x, attention_mask=mask)[0]
        out[f"{tag}.y_attn"], out[f"{tag}.y_layer"], out[f"{tag}.y_exam"] = y_attn.numpy(), y_layer.numpy(), y_exam.numpy()
        out[f"{tag}.present_k"] = present[0].numpy()
        out[f"{tag}.x"] = x.numpy()
        out[f"{tag}.meta"] = np.array([hidden, heads, T, N, k, w, seed], dtype=np.int64)
        out[f"{tag}.keys"] = np.array([k_ for k_, _ in keys_shapes])
        out[f"{tag}.shapes"] = np.array([",".join(map(str, s_)) for _, s_ in keys_shapes])
        print(tag, "ok", {k_: v_.shape for k_, v_ in out.items() if k_.startswith(tag + ".y")})
    np.savez_compressed(os.path.join(os.environ.get("GOLDEN_OUT", HERE), "callers.npz"), **out)


if __name__ == "__main__":
    main()
