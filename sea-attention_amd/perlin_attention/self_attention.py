"""PerlinSelfAttention -- module-level drop-in boundary.

Reference: src/models/perlin_attention/self_attention.py:25-264.  Callers
(src/models/perlin_opt/perlin_opt.py:434-466, perlin_bert.py:524-528) hand over the projection
Linears plus either `hidden_states` or already-projected `query/key/value_layer`; OPT passes q
pre-scaled by d^-1/2 and `query.scaling` (perlin_opt.py:562-563).  Attributes read from outside:
`.attention` (-> `.performer`), `.pconfig`, `.last_loss`, `._gradient_checkpointing`,
`.checkout_last_attention_probs`.
"""
import warnings
from typing import Optional, Tuple

import torch
from torch import nn
import torch.utils.checkpoint

from .attention import PerlinAttention, PerlinAttentionOutput
from .config import PerlinAttentionConfig, get_default_config
from .lora import LoraLinear, lora_forward_linear, lora_forward_lora


class PerlinSelfAttention(nn.Module):
    def __init__(self, config, perlin_config: PerlinAttentionConfig = None):
        super().__init__()
        self.config = config
        self.pconfig = perlin_config if perlin_config is not None else get_default_config()
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.last_loss = None

        r = self.pconfig.lora_r
        self.query_lora = LoraLinear(config.hidden_size, self.all_head_size, r)
        self.key_lora = LoraLinear(config.hidden_size, self.all_head_size, r)
        self.value_lora = LoraLinear(config.hidden_size, self.all_head_size, r)
        if self.pconfig.lora_in_approx_enabled:
            self.query_lora_for_approx_score = LoraLinear(config.hidden_size, self.all_head_size, r)
            self.key_lora_for_approx_score = LoraLinear(config.hidden_size, self.all_head_size, r)
            self.query_lora_for_approx_atten = LoraLinear(config.hidden_size, self.all_head_size, r)
            self.key_lora_for_approx_atten = LoraLinear(config.hidden_size, self.all_head_size, r)
            self.value_lora_for_approx_atten = LoraLinear(config.hidden_size, self.all_head_size, r)

        self.attention = PerlinAttention(config=config, perlin_config=perlin_config)
        self._gradient_checkpointing = False
        self.checkout_last_attention_probs = False
        self.want_attention_probs = False          # per-call request of the caller (output_attentions=True)
        self.last_attention_probs = None

    def transpose_for_scores(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 4:
            return x
        assert x.ndim == 3
        x = x.view(x.size()[:-1] + (self.num_attention_heads, self.attention_head_size))
        return x.permute(0, 2, 1, 3)

    def _project(self, linear, given, hidden_states, lora, extra_loras=()):
        """linear(hidden) (or the tensor handed in) + optional LoRA variants, all as (N,H,T,d)."""
        base = given if given is not None else lora_forward_linear(linear, hidden_states)
        x_for_lora = hidden_states if hidden_states is not None else base
        main = self.transpose_for_scores(lora_forward_lora(linear, base, lora, x_for_lora, self.pconfig.lora_enabled))
        extras = [self.transpose_for_scores(lora_forward_lora(linear, base, l, hidden_states, True)) for l in extra_loras]
        return base, main, extras

    def forward(self, query: nn.Module, key: nn.Module, value: nn.Module,
                hidden_states: torch.Tensor = None, query_layer: torch.Tensor = None,
                key_layer: torch.Tensor = None, value_layer: torch.Tensor = None,
                attention_mask: Optional[torch.FloatTensor] = None,
                attention_scores_truth: Optional[torch.FloatTensor] = None,
                context_layer_truth: Optional[torch.FloatTensor] = None,
                last_state: object = None) -> Tuple[torch.Tensor]:
        pc = self.pconfig
        if pc.layerwise and self.training:
            if hidden_states is not None:
                hidden_states = hidden_states.detach()
            else:
                assert query_layer is not None
        approx = pc.lora_in_approx_enabled

        _, key_layer, ex = self._project(
            key, key_layer, hidden_states, self.key_lora,
            (self.key_lora_for_approx_atten, self.key_lora_for_approx_score) if approx else ())
        key_layer_for_atten, key_layer_for_score = ex if approx else (key_layer, key_layer)

        _, value_layer, ex = self._project(
            value, value_layer, hidden_states, self.value_lora,
            (self.value_lora_for_approx_atten,) if approx else ())
        value_layer_for_atten = ex[0] if approx else value_layer

        q_given = query_layer is not None
        q_base = query_layer if q_given else lora_forward_linear(query, hidden_states)
        rescale = pc.causal and pc.lora_enabled and q_given
        if rescale:
            warnings.warn("causal opt does not use scaling in attention operator. it applied in query")
            q_base = q_base / query.scaling
        _, query_layer, ex = self._project(
            query, q_base, hidden_states, self.query_lora,
            (self.query_lora_for_approx_atten, self.query_lora_for_approx_score) if approx else ())
        if rescale:
            query_layer = query_layer * query.scaling
        query_layer_for_atten, query_layer_for_score = ex if approx else (query_layer, query_layer)

        args = (query_layer, key_layer, value_layer,
                query_layer_for_atten, key_layer_for_atten, value_layer_for_atten,
                query_layer_for_score, key_layer_for_score,
                attention_mask, attention_scores_truth, context_layer_truth, last_state)
        # sparse mode computes the CSR probabilities only on request.  For THIS call the module's flag is the OR of what was
        # set on the module directly, `checkout_last_attention_probs`, and what this call's caller asked for
        # (`want_attention_probs`: the OPT block's output_attentions=True); it is put back afterwards, so switching
        # `checkout_last_attention_probs` off again also stops the extra per-entry store
        att = self.attention
        want = bool(att.return_attention_probs or self.checkout_last_attention_probs or self.want_attention_probs)

        def run(*a):
            # the flag is set and restored INSIDE the function that runs the attention: a checkpointed backward calls `run`
            # again after this forward has returned, and must see the very flag the forward saw (same outputs, same
            # per-entry stores) whatever the module's attribute says by then
            before = att.return_attention_probs
            att.return_attention_probs = want
            try:
                return tuple(self.attention(*a))
            finally:
                att.return_attention_probs = before
        if self._gradient_checkpointing and self.training:
            output = PerlinAttentionOutput(*torch.utils.checkpoint.checkpoint(
                run, *args, use_reentrant=True, preserve_rng_state=True))
        else:
            output = PerlinAttentionOutput(*run(*args))

        if self.checkout_last_attention_probs:
            self.last_attention_probs = output.partial_attention_probs
        # NOTE the reference calls `output._replace(...detach())` here without keeping the result
        # (self_attention.py:260-262), i.e. nothing is detached; that effective behaviour is kept.
        self.last_loss = output.loss
        return output
