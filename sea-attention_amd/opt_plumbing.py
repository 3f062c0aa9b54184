"""Caller-side plumbing of the SEA layer for the causal (OPT) model family -- SURVEY §8a row P, BASELINE config 0.

The reference's trainer (src/trainer/perlin_trainer.py:41-155) turns command-line flags into a
`PerlinAttentionConfig`, registers it as the process default, and the model's attention block
(src/models/perlin_opt/perlin_opt.py:175-239,403-466,483-596) then builds a `PerlinSelfAttention` from that
default, forcing `causal=True` and `k_flatten_dim='causal_batch'`.  Neither the trainer nor the HF-derived model
file is part of this build; what IS the boundary of the hot path is restated here so that the same command line
reaches the same layer:

    flags --add/parse_perlin_model_options--> kwargs --perlin_config_from_options--> PerlinAttentionConfig
          --register_default_config--> SeaOPTAttention(embed_dim, num_heads) --> PerlinSelfAttention

`SeaOPTAttention` is the attention block of one decoder layer (q/k/v/out projections around the SEA layer,
q pre-scaled by d^-1/2, kv-cache tuple `(k, v[, state])`), enough to run the OPT-125m-shaped layer of BASELINE
config 0 end to end.  The compute is the HIP path; there is no CPU execution of the layer (constructing and
parameter-checking the module works anywhere, `forward` needs the GPU library).
"""
import argparse
import warnings
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
from torch import nn

from .perlin_attention import PerlinAttentionConfig, PerlinSelfAttention, get_default_config, register_default_config
from .perlin_attention.decode import SessionState
from .perlin_attention.lora import LoraLinear, lora_forward

# flag, argparse kwargs, keyword the trainer constructor receives (perlin_trainer.py:41-87); `None` defaults are
# filled from add_perlin_model_options' arguments
_FLAGS = (
    ("--method",                       dict(default="perlin", type=str),            "attention_method"),
    ("--layerwise",                    dict(action="store_true", default=False),    "perlin_layerwise"),
    ("--enable-lora",                  dict(action="store_true", default=False),    "perlin_lora"),
    ("--k",                            dict(default=None, type=int),                "perlin_k"),
    ("--k-colwise",                    dict(action="store_true", default=False),    None),          # -> not perlin_k_flatten
    ("--k-flatten-dim",                dict(default="batch", type=str),             "perlin_k_flatten_dim"),
    ("--attention-predictor-method",   dict(default="mlp", type=str),               "perlin_attention_predictor_method"),
    ("--performer-nb-feature-factor",  dict(default=None, type=float),              "perlin_performer_nb_feature_factor"),
    ("--random-lookup",                dict(action="store_true", default=False),    "perlin_random_lookup"),
    ("--random-lookup-count",          dict(default=3, type=int),                   "perlin_random_lookup_count"),
    ("--token-merging",                dict(action="store_true", default=False),    "perlin_token_merging"),
    ("--token-merging-preserve",       dict(default=0.2, type=float),               "perlin_token_merging_preserve"),
    ("--token-merging-ratio",          dict(default=0.5, type=float),               "perlin_token_merging_ratio"),
    ("--predictor-length",             dict(default=None, type=int),                "perlin_predictor_length"),
    ("--predictor-backend",            dict(default="performer", type=str),         "perlin_predictor_backend"),
    ("--n-hashs",                      dict(default=8, type=int),                   "perlin_n_hashs"),
    ("--enc-per-layer",                dict(action="store_true", default=None),     "perlin_enc_per_layer"),
    ("--context-output-method",        dict(default=None, type=str),                "perlin_context_output_method"),
    ("--k-oversample",                 dict(default=1, type=float),                 "perlin_k_oversample"),
)


def _dest(flag: str) -> str:
    return flag.lstrip("-").replace("-", "_")


def add_perlin_model_options(parser: argparse.ArgumentParser, context_output_method="norm", predictor_length=128,
                             k=7, nbf=1.0, epl=False) -> argparse.ArgumentParser:
    """Same flags, types and defaults as the reference's function of this name (perlin_trainer.py:41-62).  The OPT
    entry point calls it with `context_output_method='mix', predictor_length=256, k=64, nbf=8` style overrides."""
    late = {"--k": k, "--performer-nb-feature-factor": nbf, "--predictor-length": predictor_length,
            "--enc-per-layer": epl, "--context-output-method": context_output_method}
    for flag, kw, _ in _FLAGS:
        kw = dict(kw)
        if flag in late:
            kw["default"] = late[flag]
        parser.add_argument(flag, **kw)
    return parser


def parse_perlin_model_options(args) -> dict:
    """Namespace -> the keyword arguments of the trainer constructor (perlin_trainer.py:64-87)."""
    out = {kwarg: getattr(args, _dest(flag)) for flag, _, kwarg in _FLAGS if kwarg is not None}
    out["perlin_k_flatten"] = not args.k_colwise
    return out


def perlin_config_from_options(perlin_k=7, perlin_k_flatten=True, perlin_k_flatten_dim="batch", perlin_layerwise=False,
                               perlin_lora=False, attention_method="perlin", perlin_attention_predictor_method="mlp",
                               perlin_performer_nb_feature_factor=1, perlin_random_lookup=False,
                               perlin_random_lookup_count=3, perlin_token_merging=False,
                               perlin_token_merging_preserve=0.2, perlin_token_merging_ratio=0.5,
                               perlin_predictor_length=128, perlin_predictor_backend="performer", perlin_n_hashs=8,
                               perlin_enc_per_layer=False, perlin_context_output_method="mix", perlin_k_oversample=1,
                               compile=False, register=True, **_ignored) -> PerlinAttentionConfig:
    """What `BaseTrainer.__init__` does with those keywords (perlin_trainer.py:89-155): build the config and make it
    the process default the attention modules are constructed from.  Token merging and the non-perlin methods are
    trainer features outside the hot path; their keywords are accepted and ignored here."""
    if attention_method != "perlin":
        raise NotImplementedError(f"attention method {attention_method!r}: only the SEA ('perlin') layer is built")
    cfg = PerlinAttentionConfig(
        reformer_n_hashs=perlin_n_hashs, performer_nb_factor=perlin_performer_nb_feature_factor, k=perlin_k,
        k_flatten=perlin_k_flatten, k_flatten_dim=perlin_k_flatten_dim, random_lookup=perlin_random_lookup,
        random_lookup_count=perlin_random_lookup_count, attention_predictor_method=perlin_attention_predictor_method,
        attention_predictor_length=perlin_predictor_length, attention_predictor_backend=perlin_predictor_backend,
        attention_predictor_enc_per_layer=perlin_enc_per_layer, layerwise=perlin_layerwise, lora_enabled=perlin_lora,
        compile=compile, context_output_method=perlin_context_output_method, k_oversample=perlin_k_oversample)
    if register:
        register_default_config(cfg)
    return cfg


class SeaOPTAttention(nn.Module):
    """Attention block of one OPT decoder layer with the SEA layer inside (role of `OPTAttention` with
    `attention_method='perlin'`, perlin_opt.py:128-239,420-466,483-596).

    Parameter names follow the reference so a fine-tuned checkpoint's `...self_attn.*` entries load unchanged:
    `q_proj, k_proj, v_proj, out_proj, perlin_self_attention.*` (+ `perlin_out_lora` with LoRA enabled).
    """

    def __init__(self, embed_dim: int, num_heads: int, bias: bool = True, max_position_embeddings: int = 2048,
                 is_decoder: bool = True):
        super().__init__()
        assert embed_dim % num_heads == 0, (embed_dim, num_heads)
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.is_decoder = is_decoder
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.v_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)

        pconfig = get_default_config()
        pconfig.causal = True                                    # perlin_opt.py:224
        if pconfig.k_flatten_dim == "batch":                     # :225-227
            warnings.warn("k_flatten_dim 'batch' is not causal; the causal attention block uses 'causal_batch'")
            pconfig.k_flatten_dim = "causal_batch"
        pconfig.check_validity()
        self.pconfig = pconfig
        if pconfig.lora_enabled:
            self.perlin_out_lora = LoraLinear(embed_dim, embed_dim, pconfig.lora_r)
        self.perlin_self_attention = PerlinSelfAttention(
            SimpleNamespace(hidden_size=embed_dim, num_attention_heads=num_heads,
                            max_position_embeddings=max_position_embeddings), perlin_config=pconfig)
        self.teacher_attention_scores = None
        self.teacher_context_layer = None
        self.last_loss = None
        self.last_perlin_output = None
        self.checkout_perlin_output = False
        self._benchmarking = False
        # opt-in fast generation: one-token calls that continue a cache run as a replayed HIP graph with room for this many
        # tokens per sequence (None: every call takes the regular cached forward)
        self.decode_graph_capacity: Optional[int] = None
        self._decode_session = None

    # the reference flips `benchmarking` on every attention module of the model to select the sparse path
    @property
    def benchmarking(self) -> bool:
        return self._benchmarking

    @benchmarking.setter
    def benchmarking(self, on: bool):
        self._benchmarking = bool(on)
        self.perlin_self_attention.attention.benchmarking = bool(on)

    def _heads(self, x: torch.Tensor) -> torch.Tensor:
        n, t, _ = x.shape
        return x.view(n, t, self.num_heads, self.head_dim).transpose(1, 2).contiguous()

    def _graph_decode_step(self, q, k_new, v_new, past_key_value, output_attentions):
        """Opt-in (`decode_graph_capacity` tokens): a ONE-token call that continues a cached sequence runs as a replayed
        HIP graph (perlin_attention/decode.py).  The first such call after a prefill builds the session from the tuple's
        K / V and `PerlinAttentionState`; later calls must bring back the ticket the previous one returned.  Returns the
        forward's triple, or None when the call is not a step this path serves (the caller falls back)."""
        cap = self.decode_graph_capacity
        if (not cap or len(past_key_value) < 3 or q.shape[2] != 1 or self.training or output_attentions
                or not self.is_decoder or self.pconfig.lora_enabled or self.teacher_context_layer is not None):
            return None
        state, sess = past_key_value[2], self._decode_session
        if isinstance(state, SessionState):
            if not (state.session is sess and state.current):
                return None                                   # a stale / foreign ticket: materialize() decides in the caller
        else:
            try:
                sess = self.perlin_self_attention.attention.decode_session(state, past_key_value[0], past_key_value[1], cap)
            except AssertionError:                            # shapes / dtype / prefix the session does not cover
                return None
            self._decode_session = sess
        if sess.length >= sess.capacity:
            return None
        ctx = sess.step(q, k_new, v_new)
        L = sess.length
        dtype = self.q_proj.weight.dtype
        y = self.out_proj(ctx if ctx.dtype == dtype else ctx.to(dtype))
        self.last_loss = 0
        return y, None, (sess.k_cache[:, :, :L], sess.v_cache[:, :, :L], SessionState(sess))

    def forward(self, hidden_states: torch.Tensor, key_value_states=None,
                past_key_value: Optional[Tuple[torch.Tensor, ...]] = None, attention_mask: Optional[torch.Tensor] = None,
                layer_head_mask=None, output_attentions: bool = False, use_cache: bool = False):
        """hidden_states (N, T_new, E); attention_mask (N, 1, T_new, T_src) additive.  Returns
        (attn_output (N, T_new, E), attention probabilities or None, (k, v[, SEA state]))."""
        assert key_value_states is None, "the SEA layer is a self-attention"
        assert attention_mask is not None and layer_head_mask is None
        dtype = self.q_proj.weight.dtype
        if hidden_states.dtype != dtype:
            hidden_states = hidden_states.to(dtype)
        if attention_mask.dtype != dtype:
            attention_mask = attention_mask.clamp_min(torch.finfo(dtype).min).to(dtype)

        q = self._heads(self.q_proj(hidden_states) * self.scaling)
        self.q_proj.scaling = torch.tensor(self.scaling, dtype=dtype, device=hidden_states.device)   # read back by the LoRA path
        k, v = self._heads(self.k_proj(hidden_states)), self._heads(self.v_proj(hidden_states))
        state = None
        if past_key_value is not None:
            fast = self._graph_decode_step(q, k, v, past_key_value, output_attentions)
            if fast is not None:
                return fast
            state = past_key_value[2] if len(past_key_value) > 2 else None
            if isinstance(state, SessionState):              # a call the session cannot serve: continue from a real state
                state = state.materialize()
            k = torch.cat([past_key_value[0], k], dim=2)
            v = torch.cat([past_key_value[1], v], dim=2)
        present = (k, v) if self.is_decoder else None

        truth = self.teacher_context_layer
        if truth is not None:
            N, H, T, D = q.shape
            shape = (lambda c: c.view(N, H, T, D).transpose(1, 2).reshape(N, T, H * D))
            truth = (lambda f=truth: shape(f())) if callable(truth) else shape(truth)
        # sparse mode produces the per-entry probabilities only when somebody reads them (attention.return_attention_probs):
        # output_attentions=True and the checkout hook do, for this call only (the reference always returns them,
        # attention.py:1162-1171)
        psa = self.perlin_self_attention
        psa.want_attention_probs = bool(output_attentions) or bool(self.checkout_perlin_output)
        try:
            out = psa(
                query=self.q_proj, key=self.k_proj, value=self.v_proj, hidden_states=None, query_layer=q, key_layer=k,
                value_layer=v, attention_mask=attention_mask, attention_scores_truth=self.teacher_attention_scores,
                context_layer_truth=truth, last_state=state)
        finally:
            psa.want_attention_probs = False
        self.last_loss = out.loss
        if self.checkout_perlin_output and not self.benchmarking:
            self.last_perlin_output = out
        if out.state is not None and present is not None:
            present = (*present, out.state)

        ctx = out.context_layer
        if ctx.dtype != dtype:
            ctx = ctx.to(dtype)
        if self.pconfig.lora_enabled:
            y = lora_forward(self.out_proj, self.perlin_out_lora, ctx, True)
        else:
            y = self.out_proj(ctx)
        return y, (out.partial_attention_probs if output_attentions else None), present


class SeaOPTDecoderLayer(nn.Module):
    """One OPT decoder layer around `SeaOPTAttention` (role of `OPTDecoderLayer`, perlin_opt.py:638-800: pre- or post-norm
    self-attention + feed-forward, same sub-module names so the reference layer's state dict loads as it is).  It exists for
    the reference's measurement protocol -- `benchmark_bert.exam` (src/main/benchmark_bert.py:162-239) times
    `decoder.layers[0]` with `fc1` / `fc2` / `out_proj` replaced by `nn.Identity` -- and for the drop-in check against the
    reference's own layer (tests/test_reference_callers.py).  Dropout is the reference's (`p` applied in training only)."""

    def __init__(self, hidden_size: int, num_heads: int, ffn_dim: int, max_position_embeddings: int = 2048, bias: bool = True,
                 do_layer_norm_before: bool = True, activation=None, dropout: float = 0.0, layer_norm_elementwise_affine=True):
        super().__init__()
        self.embed_dim = hidden_size
        self.self_attn = SeaOPTAttention(hidden_size, num_heads, bias=bias, max_position_embeddings=max_position_embeddings)
        self.do_layer_norm_before = do_layer_norm_before
        self.dropout = dropout
        self.activation_fn = activation or nn.ReLU()                 # OPT's activation_function = "relu"
        self.self_attn_layer_norm = nn.LayerNorm(hidden_size, elementwise_affine=layer_norm_elementwise_affine)
        self.fc1 = nn.Linear(hidden_size, ffn_dim, bias=bias)
        self.fc2 = nn.Linear(ffn_dim, hidden_size, bias=bias)
        self.final_layer_norm = nn.LayerNorm(hidden_size, elementwise_affine=layer_norm_elementwise_affine)

    def exam_surgery(self, benchmarking: bool = True):
        """benchmark_bert.py:162-203 on this layer: the sparse mode on every module that has the switch, `fc1` / `fc2` /
        `out_proj` -> Identity (the layer then costs its attention, two LayerNorms and the residual adds)."""
        for m in self.modules():
            if hasattr(m, "benchmarking"):
                m.benchmarking = benchmarking
        self.fc1, self.fc2, self.self_attn.out_proj = nn.Identity(), nn.Identity(), nn.Identity()
        return self

    def forward(self, hidden_states, attention_mask=None, layer_head_mask=None, past_key_value=None,
                output_attentions=False, use_cache=False):
        dtype = self.self_attn.q_proj.weight.dtype
        if hidden_states.dtype != dtype:
            hidden_states = hidden_states.to(dtype)
        residual = hidden_states
        if self.do_layer_norm_before:
            hidden_states = self.self_attn_layer_norm(hidden_states)
        hidden_states, attn_weights, present = self.self_attn(
            hidden_states=hidden_states, past_key_value=past_key_value, attention_mask=attention_mask,
            layer_head_mask=layer_head_mask, output_attentions=output_attentions, use_cache=use_cache)
        hidden_states = residual + nn.functional.dropout(hidden_states, p=self.dropout, training=self.training)
        if not self.do_layer_norm_before:
            hidden_states = self.self_attn_layer_norm(hidden_states)
        shape = hidden_states.shape
        hidden_states = hidden_states.reshape(-1, shape[-1])
        residual = hidden_states
        if self.do_layer_norm_before:
            hidden_states = self.final_layer_norm(hidden_states)
        hidden_states = self.fc2(self.activation_fn(self.fc1(hidden_states)))
        hidden_states = nn.functional.dropout(hidden_states, p=self.dropout, training=self.training)
        hidden_states = (residual + hidden_states).view(shape)
        if not self.do_layer_norm_before:
            hidden_states = self.final_layer_norm(hidden_states)
        out = (hidden_states,)
        if output_attentions:
            out += (attn_weights,)
        if use_cache:
            out += (present,)
        return out


def causal_additive_mask(N: int, T_dst: int, T_src: int, dtype, device) -> torch.Tensor:
    """(N, 1, T_dst, T_src) mask as the OPT decoder prepares it: 0 where key s may be seen by query t
    (s <= T_src - T_dst + t), the dtype's lowest value elsewhere."""
    t = torch.arange(T_dst, device=device).view(-1, 1) + (T_src - T_dst)
    s = torch.arange(T_src, device=device).view(1, -1)
    m = torch.zeros(T_dst, T_src, dtype=dtype, device=device).masked_fill_(s > t, torch.finfo(dtype).min)
    return m.view(1, 1, T_dst, T_src).expand(N, 1, T_dst, T_src)


def main(argv=None):
    """BASELINE config 0 in miniature: the reference's OPT flags -> config -> one OPT-125m-shaped SEA attention block
    -> one forward over a 2048-token batch of synthetic hidden states on cuda:0."""
    p = argparse.ArgumentParser(description=main.__doc__)
    add_perlin_model_options(p, context_output_method="mix", predictor_length=256, k=64, nbf=8)
    p.add_argument("--hidden", type=int, default=768)
    p.add_argument("--heads", type=int, default=12)
    p.add_argument("--seq-len", type=int, default=2048)
    p.add_argument("--batch", type=int, default=1)
    p.add_argument("--dtype", default="bf16", choices=("bf16", "fp16", "fp32"))
    args = p.parse_args(argv)
    cfg = perlin_config_from_options(**parse_perlin_model_options(args))
    print(cfg)
    torch.manual_seed(42)
    block = SeaOPTAttention(args.hidden, args.heads, max_position_embeddings=args.seq_len)
    if not torch.cuda.is_available():
        raise SystemExit("config and module construction OK; the forward needs the MI355X (no CPU path)")
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    block = block.to("cuda", dt).eval()
    block.benchmarking = True
    x = torch.randn(args.batch, args.seq_len, args.hidden, device="cuda", dtype=dt)
    mask = causal_additive_mask(args.batch, args.seq_len, args.seq_len, dt, "cuda")
    with torch.no_grad():
        y, _, present = block(x, attention_mask=mask)
    torch.cuda.synchronize()
    print("output", tuple(y.shape), y.dtype, "finite" if torch.isfinite(y).all() else "NOT FINITE",
          "kv", tuple(present[0].shape))


if __name__ == "__main__":
    main()
