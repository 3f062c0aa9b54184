"""PerlinAttention -- the SEA attention module, MI355X build.

Drop-in for the reference class of the same name (src/models/perlin_attention/attention.py:133-1359):
same constructor, same `forward` signature (:333-347), same parameter / buffer names (trained
checkpoints load), same `benchmarking` mode switch (:152), same `PerlinAttentionOutput` (:84-106),
same timing-region and temp-buffer names (they are the reference's parity probes,
src/main/tests/test_perlin_opt_consist.py:198-232).

Two modes, as in the reference:

* `benchmarking = False` (default; training / PPL evaluation): dense T x T scores with an additive
  mask (:1061-1133).  Plain torch, runs on CPU too -- this is the reference's own CPU-runnable path.
* `benchmarking = True`: the sparse path.  Steps H..K of SURVEY.md section 3B run as hand-written HIP
  kernels through libsea_hip.so:
      grouped top-k + interpolation -> flat CSR      (ops.topk_to_csr;   replaces :774-947, :1034-1042)
      SDDMM + softmax + scale + SpMM + avg-pool mix   (ops.sparse_attention; replaces :1158-1173, :1236-1237)
  with no host synchronisation between them.  There is no fallback: without the library or on a CPU
  tensor this mode raises.

Scope of this round: the causal configuration (`pconfig.causal=True`, `k_flatten_dim='causal_batch'`,
the only one the OPT/LLaMA path constructs -- perlin_opt.py:224-239) without kv-cache.
"""
import math
import os
import warnings
from typing import NamedTuple, Optional

import torch
import torch.nn.functional as F
from torch import nn

from ..utils import get_bench
from .config import PerlinAttentionConfig, get_default_config
from .modules import CausalConv2d, KeepRes, UpsampleFP32
from .performer import FastAttention, ProjectionUpdater
from . import ops

timer = lambda name: get_bench().region(name)
mem = lambda name: get_bench().mem_region(name)
_C8_DECLINED = set()      # (dtype, Cin, Cout, k) of predictor convolutions the hand-written kernels declined: warned about once each


def _safe_to(t, d):
    return t.to(d) if isinstance(t, torch.Tensor) else t


class PerlinAttentionOutput(NamedTuple):
    loss: torch.Tensor
    context_layer: torch.Tensor
    partial_attention_probs: torch.Tensor
    partial_attention_mask: torch.Tensor
    estimated_attention_probs_m: torch.Tensor
    estimated_attention_probs: torch.Tensor
    dense_attention_probs: torch.Tensor
    key_for_score: torch.Tensor
    state: object

    def to(self, device):
        return PerlinAttentionOutput(*[_safe_to(f, device) for f in self])


class ModuleBenchmark(nn.Module):
    """Named timing wrapper; keeps the `.module` nesting of the reference's state dict (:108-121)."""

    def __init__(self, name, module, disabled=False):
        super().__init__()
        self.name = name
        self.module = module
        self.disabled = disabled

    def forward(self, x):
        if self.disabled:
            return self.module(x)
        with timer(self.name):
            return self.module(x)


class ChannelSplit(nn.Module):
    """(N, C, H, W) -> (N, C*split, H, W/split): the decoder row is cut into `split` conv channels (:123-131)."""

    def __init__(self, split):
        super().__init__()
        self.split = split

    def forward(self, x):
        N, C, H, W = x.shape
        s = self.split
        return x.view(N, C, H, s, W // s).permute(0, 1, 3, 2, 4).reshape(N, C * s, H, W // s)


def _kl_and_mse(est_scores, truth_scores, causal_mask, fp_min):
    """KD terms of the causal path (:741-763 / :1086-1103): 0.1*KL(batchmean) + MSE on fp32 softmaxes."""
    dead = causal_mask < -1
    W = est_scores.shape[-1]
    logp = F.log_softmax(est_scores.masked_fill(dead, fp_min).float(), dim=-1).view(-1, W)
    tgt = F.softmax(truth_scores.masked_fill(dead, fp_min).float(), dim=-1).view(-1, W)
    kl = F.kl_div(logp, tgt, reduction='batchmean') * 0.1
    mse = F.mse_loss(F.softmax(est_scores.masked_fill(dead, fp_min).float(), dim=-1).view(-1, W), tgt)
    return kl + mse


def _softmax_like_reference(x: torch.Tensor, training: bool) -> torch.Tensor:
    """`softmax_bf16` (attention.py:62-72): plain softmax at inference; in training under 16-bit autocast the softmax runs in
    fp32 and the result returns to the input dtype."""
    if not training:
        return torch.softmax(x, dim=-1)
    op_dtype = torch.float32 if torch.get_autocast_gpu_dtype() in (torch.bfloat16, torch.float16) else x.dtype
    y = torch.softmax(x.to(op_dtype), dim=-1)
    return y.to(x.dtype)


class PerlinAttention(nn.Module):
    def __init__(self, config, perlin_config: PerlinAttentionConfig = None):
        super().__init__()
        self.config = config
        self.pconfig = perlin_config if perlin_config is not None else get_default_config()

        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self._warning_messages = ""

        # mode switch, set from outside by walking model.modules() (benchmark_bert.py:172-173)
        self.benchmarking = False
        # dtype of context_layer in sparse mode; None = the reference's behaviour (fp32, sdbmm.py:347)
        self.context_layer_dtype = None
        # sparse mode: None = decide by inspecting the mask (one host sync, as the reference does at :434);
        # True/False = the caller already knows whether the batch carries padding
        self.assume_not_padded = None
        # sparse mode: return the mask as a torch.sparse_csr_tensor (int64, one host sync for Z) instead of
        # the internal int32 FlatCSR handle
        self.materialize_csr = False
        # sparse mode: also produce `partial_attention_probs` (per-entry rs * softmax on the mask's CSR, what the reference
        # returns at :1162-1171); costs one extra 4-byte store + reload per entry, so it is on only when somebody reads it:
        # PerlinSelfAttention.checkout_last_attention_probs sets it, so does probing through get_bench()
        self.return_attention_probs = False
        # sparse mode: `estimated_attention_probs(_m)` of the output as an `ops.LazyTensor` when the fused tail + selection
        # launch produced the map: the (N,H,T,T_M) tensor is then NOT written by the step (nothing on the hot path reads it;
        # 537 MB per step at OPT-1.3B x 8) and is computed from the kept conv output the first time a caller touches its
        # values (bit-identical).  False = always write it (the reference's eager tensor, attention.py:1343)
        self.lazy_attention_probs = True
        # sparse mode: the flat CSR's COLUMN array is an output nobody on the hot path reads (the fused attention launch expands
        # the kept pixels itself and walks them from LDS); True = that launch does not write it (266 MB per step at OPT-1.3B x
        # 8) and `partial_attention_mask` keeps its columns pending -- any reader of `.col` / `.col_indices()` / the wire
        # format runs the emit launch then (bit-identical).  False = written by the step
        self.lazy_csr_columns = True
        # C8 predictor CNN: True = the last (conv, ReLU) launch also evaluates the tail's 1x1 convolution on the tile it holds
        # and hands z (N, T, H, T_M/4) fp32 to the tail instead of its activation (`sea_causal_conv_c8_z`; bit-identical, the
        # activation of the last pair is then not written).  Built in round 5 and MEASURED SLOWER (conv2 +27 us, tail
        # +-0 at OPT-1.3B x 8 -- the tail's z stage is latency its other resident waves already hide): off by default
        self.conv_z_epilogue = False
        # sparse mode: kernel of steps J-L: "gather" (row-indexed gathers), "tile" (MFMA tile kernel, 16-bit data,
        # d in {64, 80, 128}; wins when neighbouring query rows keep mostly the same keys -- trained predictors), "auto"
        self.sparse_kernel = "auto"
        # HIP Performer: None = the library's plan for the shape (it cuts ONE long sequence into segments so that all CUs
        # work: another fp32 summation order than the one-pass kernel a full batch takes), 1 = always the one-pass kernel
        # (bench.py's self-check runs a batch item alone with it, so that the item reproduces its batched rows bit for bit)
        self.performer_segments = None
        # debugging / parity: run the estimator's LayerNorm / conv tail through the torch modules even on GPU
        self.force_torch_estimator = False
        # dense mode (training / evaluation): None = the reference's form, every (N,H,T,T) tensor materialised at once
        # (scores, both masks, both probability tensors: 5 x 8.6 GB in fp32 at one OPT-2.7B sequence of 8192 tokens);
        # an integer = that many heads at a time: same context and same loss (the KD terms are means over heads, summed
        # chunk by chunk), the T x T tensors live for one chunk only and the three T x T OUTPUT fields are None
        self.dense_head_chunk = None
        self._fused_gates = None            # (row_scale, average_scale) when the fused predictor MLP produced them
        self._fused_selection = None        # (bits, row_nnz, head_off) when the tail kernel also ran the top-k selection
        self._avg_ahead = None              # cumulative average of v when the Performer launch produced it

        d, H = self.attention_head_size, self.num_attention_heads
        pc = self.pconfig

        # ---- estimator: Performer -----------------------------------------------------------------
        self.performer_nb_features = int(d * math.log(d) / pc.performer_nb_factor)
        self.performer = FastAttention(dim_heads=d, nb_features=self.performer_nb_features,
                                       causal=pc.causal, generalized_attention=pc.causal)
        self.performer_proj_updater = ProjectionUpdater(self.performer, 1000)
        if pc.attention_predictor_backend != 'performer':
            # 'cosformer' (attention.py:169-179, 535-549) needs the reference's third-party cosformer module and is commented
            # "not supported" there; anything else raises in the reference's forward (:550)
            raise Exception(f"attention_predictor_backend {pc.attention_predictor_backend!r}: this build estimates with the Performer only")

        # ---- predictor: MLP encoder, row decoder, CNN ---------------------------------------------
        pv = d * 3
        self.register_buffer('attention_predictor_enc_head_embd', torch.eye(H))
        self.attention_predictor_enc_per_layer = nn.Sequential(
            nn.Linear(pv * H, d * 2 * H), nn.LayerNorm(d * 2 * H), nn.GELU())
        self.attention_predictor_enc = nn.Sequential(nn.Linear(pv, d * 2), nn.LayerNorm(d * 2), nn.GELU())
        T_M = pc.attention_predictor_length
        if not pc.causal:
            self.attention_predictor_dec_row_down_scale = 2
            self.attention_predictor_dec_row_splits = 4
            self.attention_predictor_dec_row_out_ch = (T_M // 2) * 4
            self.attention_predictor_dec_row = nn.Sequential(
                nn.Linear(d * 2, self.attention_predictor_dec_row_out_ch), ChannelSplit(4))
            self.attention_predictor_cnn = nn.Sequential(KeepRes(
                nn.Conv2d(4 * H, 4 * H, 3, padding=1, stride=(2, 1)), nn.ReLU(),
                nn.Conv2d(4 * H, 4 * H, 3, padding=1), nn.ReLU(),
                UpsampleFP32((2, 1), torch.float16),
                nn.Conv2d(4 * H, H, 3, padding=1),
                output_width=T_M))
        else:
            inner = int(os.environ.get("PERLIN_HOTFIX_OPT_INNER_CH", "2"))
            if inner != 2:
                self._warning_messages += f'WARN, you are using hotfix backend. PERLIN_HOTFIX_OPT_INNER_CH {inner}\n'
            self.attention_predictor_dec_row_down_scale = 4
            self.attention_predictor_dec_row_splits = inner
            self.attention_predictor_dec_row_out_ch = (T_M // 4) * inner
            self.attention_predictor_dec_row = nn.Sequential(
                nn.Linear(d * 2, self.attention_predictor_dec_row_out_ch), ChannelSplit(inner))
            deeper = int(os.environ.get("PERLIN_HOTFIX_OPT_DEEPER", "0")) == 1
            if deeper:
                self._warning_messages += 'WARN, you are using hotfix backend. PERLIN_HOTFIX_OPT_DEEPER\n'
            conv = lambda name: ModuleBenchmark(
                name, CausalConv2d(inner * H, inner * H, 3, padding=2, dilation=2, stride=(1, 1), causal=True))
            body = [conv('cnn.keepres.conv1'), nn.ReLU(), conv('cnn.keepres.conv2'), nn.ReLU()]
            if deeper:
                body += [conv('cnn.keepres.conv3'), nn.ReLU()]
            body += [ModuleBenchmark('cnn.keepres.upsam', UpsampleFP32((1, 4), torch.float16)),
                     ModuleBenchmark('cnn.keepres.conv4', CausalConv2d(inner * H, H, 1, padding=1, causal=True))]
            self.attention_predictor_cnn = nn.Sequential(
                ModuleBenchmark('cnn.lnorm1', nn.LayerNorm(T_M // 4)),
                ModuleBenchmark('cnn.keepres', KeepRes(*body, output_width=T_M)),
                ModuleBenchmark('cnn.lnorm2', nn.LayerNorm(T_M)))        # keeps the causal model from exploding
        self.attention_predictor_dec_scaler = nn.Sequential(nn.Linear(d * 2, 2))

        # ---- compressed predictor (attention.py:293-312): softmax over a codebook per patch, patches concatenated to the map ----
        if pc.attention_predictor_method == 'comp':
            self.attention_predictor_comp_length = pc.attention_predictor_comp_patch_count * pc.attention_predictor_comp_patch_size
            self.attention_predictor_comp_codebook = nn.Parameter(
                torch.randn((pc.attention_predictor_comp_book_size, pc.attention_predictor_comp_patch_size)))
            self.attention_predictor_comp_enc = nn.Sequential(
                nn.Dropout(0.1), nn.Linear(pv, d * 2), nn.LayerNorm(d * 2), nn.GELU())
            self.attention_predictor_comp_dec_row = nn.Sequential(
                nn.Linear(d * 2, pc.attention_predictor_comp_book_size * pc.attention_predictor_comp_patch_count))
        elif pc.attention_predictor_method != 'mlp':
            raise Exception(f"attention_predictor_method {pc.attention_predictor_method!r}")      # (attention.py:663)

        # ---- output ------------------------------------------------------------------------------
        self.norm_performer = nn.LayerNorm(config.hidden_size)
        self.norm_partial = nn.LayerNorm(config.hidden_size)
        self.norm_random = nn.LayerNorm(config.hidden_size)
        self.norm = nn.LayerNorm(config.hidden_size)
        self.register_buffer('_v_eye', None, persistent=False)
        self.v_eye_learned = nn.Parameter(torch.rand((1, 1, d, d)))
        max_pos = getattr(config, 'max_position_embeddings', 2048)
        self.v_eye_learned_causal = nn.Parameter(torch.randn((1, 1, max_pos, d)))

        self._keep_cache = {}
        # packed / re-laid-out copies of the predictor's weights are cached per tensor (ops.predictor._cached); a loaded
        # state dict or a device move may come with weight edits the cache cannot see
        self.register_load_state_dict_post_hook(lambda module, incompatible: ops.clear_prep_cache())

    def _apply(self, fn, *args, **kwargs):
        # device / dtype moves: the cache's own keys (tensor identity, address, version) already miss afterwards; the clear
        # also frees the old packs.  A train()/eval() switch edits no weight and does NOT clear.
        ops.clear_prep_cache()
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------------------------------------------
    def _fuses_interpolation(self, q, T_M) -> bool:
        """Steps I + J in one launch (`sea_sparse_attention_fused`): with `sparse_kernel` "gather" or "auto" and a head shape
        the fused gather kernels cover, the CSR's column array is left to the attention launch.  "auto" takes it wherever
        it exists: fused gather beats emit + the better of {gather, tile} on every map measured (OPT-1.3B x 8: layer's own
        0.99 + 0.03 vs 0.96 + 0.18 ms, structured 0.98 vs tile 0.90 + 0.18; LLaMA-13B: gather wins anyway); the plan-based
        choice between the two kernels remains for shapes without a fused form (rows narrower than 8 or wider than 16 lanes,
        T_M % 32 != 0, a decoding step's few rows: `ops.fused_interp_supported`; 16-bit d = 64 / 80 / 128 all have one)."""
        rows = q.shape[0] * q.shape[1] * q.shape[2]              # a decoding step's few rows take the wave-per-row kernel
        return self.sparse_kernel in ("gather", "auto") and ops.fused_interp_supported(q.dtype, q.shape[-1], T_M, rows)

    def _keep_table(self, H, T_dst, T_src, T_M, device):
        """K_t (int32, device) + analytic CSR capacity, cached per shape (host work only once)."""
        k, os_ = self.pconfig.k, self.pconfig.k_oversample
        key = (H, T_dst, T_src, T_M, k, os_, str(device))
        hit = self._keep_cache.get(key)
        if hit is None:
            keep_cpu = ops.keep_table_causal(H, T_dst, T_M, k, os_)
            z_cap = ops.z_capacity(keep_cpu, H, T_dst, T_src, T_M, int(k), True)
            hit = (keep_cpu.to(device), z_cap)
            self._keep_cache[key] = hit
        return hit

    def _hip_estimator_ok(self, x):
        """HIP kernels for ChannelSplit+LN1 / predictor tail apply when: device tensor, autograd off, the standard
        causal predictor layout, and shapes inside the kernels' limits.  Otherwise the torch modules run."""
        if self.force_torch_estimator or not x.is_cuda or torch.is_grad_enabled() or not self.pconfig.causal:
            return False
        T_M = self.pconfig.attention_predictor_length
        body = list(self.attention_predictor_cnn[1].module.net.children())
        up, conv4 = body[-2].module, body[-1].module
        vec = 4 if x.dtype == torch.float32 else 8
        return (isinstance(up, UpsampleFP32) and tuple(up.scale) == (1, 4) and isinstance(conv4, CausalConv2d)
                and conv4.kernel_size == 1 and conv4.padding == (0, 1) and conv4.causal
                and T_M % 4 == 0 and T_M <= 512 and (T_M // 4) % vec == 0
                and ((T_M // 4 + 63) // 64) * ((self.num_attention_heads + 7) // 8) <= 16)

    def _fused_mlp_ok(self, x):
        """One-launch predictor MLP (csrc/sea_mlp.hip): inference on 16-bit data with the standard
        enc = Linear+LayerNorm+GELU / dec_row = Linear+ChannelSplit(2) / 2-way gate modules, C8 CNN available.
        Dense and sparse mode share it (so both see the same probability map); dense mode also asks for the
        encoder output, which its gate / probing code reads."""
        if not (self._hip_estimator_ok(x) and x.dtype in (torch.float16, torch.bfloat16)):
            return False
        if self.pconfig.attention_predictor_enc_per_layer or int(os.environ.get('QUERY_SKIPS', '1')) != 1:
            return False
        enc, dec, sc = self.attention_predictor_enc, self.attention_predictor_dec_row, self.attention_predictor_dec_scaler
        if not (len(enc) == 3 and isinstance(enc[0], nn.Linear) and isinstance(enc[1], nn.LayerNorm)
                and isinstance(enc[2], nn.GELU) and getattr(enc[2], 'approximate', 'none') == 'none'
                and len(dec) == 2 and isinstance(dec[0], nn.Linear) and self.attention_predictor_dec_row_splits == 2
                and len(sc) == 1 and isinstance(sc[0], nn.Linear) and sc[0].out_features == 2):
            return False
        if any(m.bias is None for m in (enc[0], dec[0], sc[0])) or not enc[1].elementwise_affine:
            return False
        ln1 = self.attention_predictor_cnn[0].module
        body = list(self.attention_predictor_cnn[1].module.net.children())
        return (isinstance(ln1, nn.LayerNorm) and ln1.elementwise_affine
                and tuple(ln1.normalized_shape) == (dec[0].out_features // 2,)
                and ops.predictor_mlp_supported(enc[0].out_features, dec[0].out_features, x.shape[1], x.shape[-1])
                and self._c8_cnn_ok(x, body))

    def _c8_cnn_ok(self, x, body):
        """Channel-blocked (C8) MFMA conv pipeline: body = (CausalConv2d k x k, ReLU)* + upsample + 1x1 conv.  16-bit data on the
        bf16 / f16 MFMA kernel; fp32 data (round 5) on the fp32-MFMA kernel while its weight image fits the LDS."""
        if x.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            return False
        W = self.pconfig.attention_predictor_length // 4
        if W % 8 or W > 512 or (len(body) - 2) % 2:
            return False
        for i in range(0, len(body) - 2, 2):
            conv = getattr(body[i], 'module', None)
            if not (isinstance(conv, CausalConv2d) and conv.causal and isinstance(body[i + 1], nn.ReLU)
                    and conv.stride in (1, (1, 1)) and isinstance(conv.dilation, int)
                    and conv.dilation * (conv.kernel_size - 1) == 2 * conv.padding[1]
                    and conv.kernel_size in (1, 3)
                    and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0 and conv.out_channels <= 128):
                return False
            if x.dtype == torch.float32:
                fits = ops.conv_c8_f32_supported(conv.in_channels, conv.out_channels, conv.kernel_size)
            else:
                fits = (16 * ((conv.out_channels + 15) // 16) * (conv.kernel_size ** 2 * ((conv.in_channels + 31) // 32 * 32) * 2 + 4) <= 160 * 1024
                        and (conv.out_channels + 15) // 16 in (1, 2, 3, 4, 5, 6, 7, 8))
            if not fits:
                # not silent: the layer still runs -- on the framework's convolutions (MIOpen), several times slower
                key = (x.dtype, conv.in_channels, conv.out_channels, conv.kernel_size)
                if key not in _C8_DECLINED:
                    _C8_DECLINED.add(key)
                    warnings.warn(f"SEA predictor CNN: {conv.in_channels} -> {conv.out_channels} channels, {conv.kernel_size} x "
                                  f"{conv.kernel_size}, {x.dtype}: the weight image does not fit the 160 KB LDS of the hand-written "
                                  "convolution kernel; this layer takes the framework's convolutions (MIOpen) instead")
                return False
        return True

    def _c8_cnn_and_tail(self, x, body, ln2, want_scores, allow_select, q, T_SRC):
        """Steps F-G (+ H) on the C8 kernels: the (conv, ReLU) pairs of `cnn.keepres`, then upsample + 1x1 conv + area resize +
        `cnn.lnorm2` + softmax -- with the top-k selection in the same launch when `allow_select` (sparse mode, unpadded
        batch, nobody probing).  `conv_z_epilogue`: the LAST pair's launch also evaluates the 1x1 convolution on the tile it
        holds (`sea_causal_conv_c8_z`) and hands z to the tail instead of the activation (bit-identical; the activation of the
        last pair is then never written).  Returns (probs, scores); `self._fused_selection` is set when the selection ran."""
        conv4 = body[-1].module
        T_M_, Hh = self.pconfig.attention_predictor_length, self.num_attention_heads
        w1 = conv4.weight[:, :, 0, 0]
        pairs = list(range(0, len(body) - 2, 2))
        last = body[pairs[-1]].module if pairs else None
        use_z = (self.conv_z_epilogue and last is not None and x.dtype in (torch.float16, torch.bfloat16)
                 and ops.conv_z_supported(last.out_channels, Hh, last.kernel_size, x.shape[3]))
        z = None
        with timer("cnn.keepres"):
            for li_ in pairs:                                                               # (conv, ReLU) pairs
                conv = body[li_].module
                with timer(body[li_].name):
                    if use_z and li_ == pairs[-1]:
                        _y, z = ops.causal_conv_c8_z(x, conv.weight, conv.bias, conv.kernel_size, conv.dilation, conv.padding[1],
                                                     w1, conv4.bias, ln2.weight, ln2.bias, relu=True)
                    else:
                        x = ops.causal_conv_c8(x, conv.weight, conv.bias, conv.kernel_size, conv.dilation,
                                               conv.padding[1], relu=True)
            with timer("cnn.tail"):
                if allow_select and ops.predictor_tail_select_supported(x, Hh, T_M_):
                    # sparse mode, unpadded batch: the tail hands the map to the top-k selection in registers (one launch)
                    keep, _z = self._keep_table(Hh, q.shape[-2], T_SRC, T_M_, q.device)
                    probs, scores, sel = ops.predictor_tail_select(
                        None if z is not None else x, w1, conv4.bias, ln2.weight, ln2.bias, up=4, T_m=T_M_,
                        keep=keep, k=int(self.pconfig.k), T_src=T_SRC, is_causal=True, eps=ln2.eps,
                        want_scores=want_scores, lazy_probs=self.lazy_attention_probs, z=z, map_dtype=x.dtype)
                    self._fused_selection = (probs, sel)
                elif z is not None:
                    probs, scores = ops.predictor_tail_z(z, w1, conv4.bias, ln2.weight, ln2.bias, up=4, T_m=T_M_,
                                                         dtype=x.dtype, eps=ln2.eps, want_scores=want_scores)
                else:
                    probs, scores = ops.predictor_tail(x, w1, conv4.bias, ln2.weight, ln2.bias, up=4,
                                                       T_m=T_M_, eps=ln2.eps, want_scores=want_scores)
        return probs, scores

    def _estimate(self, q, k, v, q_for_atten, k_for_atten, v_for_atten, dst_attention_mask, not_padded, T_SRC):
        """Steps A..G: value augmentation, Performer, predictor MLP + CNN, softmax over T_M."""
        bench = get_bench()
        self._avg_ahead = None
        hip_perf = (self._hip_estimator_ok(q) and not_padded and q_for_atten.shape == v_for_atten.shape
                    and ops.performer_supported(q.shape[-1], self.performer_nb_features)
                    and T_SRC == q.shape[-2] and q_for_atten.dtype == k_for_atten.dtype == v_for_atten.dtype)
        if hip_perf:
            # value augmentation + Performer + concat with v: one fp32-MFMA kernel (csrc/sea_performer.hip)
            with timer("performer"):
                pos = self.v_eye_learned_causal[0, 0, :T_SRC, :]
                # sparse mode: the same launch also emits the cumulative average of v that step K mixes in
                # (only when the estimator's value tensor IS the layer's value tensor, i.e. no separate LoRA branch)
                avg_too = (self.benchmarking and v_for_atten is v and ops.performer_avg_supported(q_for_atten, self.performer_nb_features))
                performer_value = ops.performer_value(q_for_atten, k_for_atten, v_for_atten, pos,
                                                      self.performer.projection_matrix, want_avg=avg_too,
                                                      n_segments=self.performer_segments)
                if avg_too:
                    performer_value, self._avg_ahead = performer_value
                D_ = q.shape[-1]
                bench.register_temp_buffer('v_for_atten', None, lazy=lambda: torch.cat(
                    [pos.expand(v_for_atten.shape).to(v_for_atten.dtype), v_for_atten], dim=-1))
                bench.register_temp_buffer('performer_context_layer', performer_value[..., :2 * D_])
                bench.register_temp_buffer('performer_value', performer_value)
        else:
            with timer("vmask"):
                with timer("vmask.cat_fill"):
                    pos = self.v_eye_learned_causal[:, :, :T_SRC, :]
                    v_for_atten = torch.cat([pos.expand(v_for_atten.shape).to(v_for_atten.dtype), v_for_atten], dim=-1)
                    bench.register_temp_buffer('v_for_atten', v_for_atten)
                    if not not_padded:
                        v_for_atten = v_for_atten.masked_fill(dst_attention_mask < -1, 0)
                        v = v.masked_fill(dst_attention_mask < -1, 0)
            with timer("performer"):
                # the estimator always runs in fp32 (attention.py:520-534)
                performer_context_layer = self.performer(q_for_atten.float(), k_for_atten.float(), v_for_atten.float())
                performer_context_layer = performer_context_layer.to(q_for_atten.dtype)
                bench.register_temp_buffer('performer_context_layer', performer_context_layer)
            with timer("performer_value"):
                performer_value = torch.cat([performer_context_layer, v], dim=-1)
                bench.register_temp_buffer('performer_value', performer_value)
        self._fused_gates = None
        self._fused_selection = None
        if self.pconfig.attention_predictor_method == 'comp':
            # attention.py:649-661: the small MLP and the codebook product stay torch modules (library GEMMs); the map's width is
            # patch_count * patch_size, whatever attention_predictor_length says; steps H..L below are the HIP kernels as always
            with timer("predictor"):
                N, H, T, _ = performer_value.shape
                pc = self.pconfig
                t_attention_predictor = self.attention_predictor_comp_enc(performer_value)
                score = self.attention_predictor_comp_dec_row(t_attention_predictor)
                score = score.view(N, H, T, pc.attention_predictor_comp_patch_count, pc.attention_predictor_comp_book_size)
                score = _softmax_like_reference(score, self.training)
                score = torch.matmul(score.view(-1, pc.attention_predictor_comp_book_size),
                                     self.attention_predictor_comp_codebook.to(score.dtype))
                estimated_attention_score = score.view(N, H, T, -1)
                bench.register_temp_buffer('t_attention_predictor', t_attention_predictor)
            with timer("mask_softmax"):
                estimated_attention_probs = _softmax_like_reference(estimated_attention_score, self.training)
            bench.register_temp_buffer('estimated_attention_score', estimated_attention_score)
            bench.register_temp_buffer('estimated_attention_probs', estimated_attention_probs)
            return v, t_attention_predictor, estimated_attention_score, estimated_attention_probs
        fused_mlp = self._fused_mlp_ok(performer_value)
        if fused_mlp:
            with timer("predictor"):
                want_scores = get_bench().activate_temp_buffers or (not self.benchmarking)
                with timer("predictor.mlp"):   # enc + dec_row + lnorm1 (+ the two gates) in one launch
                    x, t_attention_predictor, row_scale, avg_scale = ops.predictor_mlp(
                        performer_value, self.attention_predictor_enc[0], self.attention_predictor_enc[1],
                        self.attention_predictor_dec_row[0], self.attention_predictor_cnn[0].module,
                        self.attention_predictor_dec_scaler[0], want_tpred=want_scores)
                    if self.benchmarking:
                        self._fused_gates = (row_scale, avg_scale)
                with timer("predictor.cnn"):
                    keepres, ln2 = self.attention_predictor_cnn[1].module, self.attention_predictor_cnn[2].module
                    estimated_attention_probs, estimated_attention_score = self._c8_cnn_and_tail(
                        x, list(keepres.net.children()), ln2, want_scores,
                        self.benchmarking and not_padded and not get_bench().activate_temp_buffers, q, T_SRC)
                bench.register_temp_buffer('t_attention_predictor', t_attention_predictor)
            bench.register_temp_buffer('estimated_attention_score', estimated_attention_score)
            bench.register_temp_buffer('estimated_attention_probs', estimated_attention_probs)
            return v, t_attention_predictor, estimated_attention_score, estimated_attention_probs
        with timer("predictor"):
            query_skips = int(os.environ.get('QUERY_SKIPS', '1'))
            with timer("predictor.enc"):
                if self.pconfig.attention_predictor_enc_per_layer:
                    N, H, T, D3 = performer_value.shape
                    x = performer_value.permute(0, 2, 1, 3).reshape(N, T, H * D3)
                    t_attention_predictor = self.attention_predictor_enc_per_layer(x)
                    t_attention_predictor = t_attention_predictor.view(N, T, H, -1).permute(0, 2, 1, 3)
                else:
                    x = performer_value
                    if query_skips > 1:
                        assert (x.shape[-2] % query_skips) == 0
                        x = x[:, :, ::query_skips, :]
                    if self._hip_estimator_ok(x) and x.shape[-1] * 2 // 3 <= (256 if x.dtype == torch.float32 else 512):
                        enc = self.attention_predictor_enc                      # Linear -> [LayerNorm + GELU fused in HIP]
                        t_attention_predictor = ops.split_layernorm(enc[0](x), 1, enc[1].weight, enc[1].bias,
                                                                    enc[1].eps, gelu=True)
                    else:
                        t_attention_predictor = self.attention_predictor_enc(x)
            # HIP fast path for the bandwidth-bound estimator pieces: device tensors, no autograd, and the
            # standard causal predictor layout (ChannelSplit+LN, ..., upsample x4, 1x1 conv pad 1, LN, softmax)
            use_hip = self._hip_estimator_ok(t_attention_predictor)
            want_scores = get_bench().activate_temp_buffers or (not self.benchmarking)
            if use_hip:
                cnn = self.attention_predictor_cnn
                ln1, keepres, ln2 = cnn[0].module, cnn[1].module, cnn[2].module
                with timer("predictor.dec_row"):
                    dec = self.attention_predictor_dec_row[0](t_attention_predictor)          # Linear only
                    bench.register_temp_buffer('estimated_attention_score_dec_row', None,
                                               lazy=lambda: self.attention_predictor_dec_row[1](dec))
                with timer("predictor.cnn"):
                    body = list(keepres.net.children())
                    c8 = self._c8_cnn_ok(dec, body)
                    with timer("cnn.lnorm1"):
                        if c8 and dec.dtype != torch.float32:   # the CNN runs channel-blocked (C8) on the hand-written MFMA conv kernels
                            x = ops.split_layernorm_c8(dec, self.attention_predictor_dec_row_splits, ln1.weight, ln1.bias, ln1.eps)
                        elif c8:    # fp32 data: LayerNorm kernel + one re-layout pass (35 + 45 us at config 2; the one-launch
                            #         split_layernorm_c8 takes fp32 too but its LDS transpose is 8-way bank-conflicted there: 105 us)
                            x = ops.to_c8(ops.split_layernorm(dec, self.attention_predictor_dec_row_splits, ln1.weight, ln1.bias, ln1.eps))
                        else:
                            x = ops.split_layernorm(dec, self.attention_predictor_dec_row_splits, ln1.weight, ln1.bias, ln1.eps)
                    if c8:
                        estimated_attention_probs, estimated_attention_score = self._c8_cnn_and_tail(
                            x, body, ln2, want_scores,
                            query_skips == 1 and self.benchmarking and not_padded and not get_bench().activate_temp_buffers, q, T_SRC)
                    else:
                        with timer("cnn.keepres"):
                            for layer in body[:-2]:                                           # causal convs + ReLUs
                                x = layer(x)
                            conv4 = body[-1].module
                            with timer("cnn.tail"):
                                estimated_attention_probs, estimated_attention_score = ops.predictor_tail(
                                    x, conv4.weight[:, :, 0, 0], conv4.bias, ln2.weight, ln2.bias, up=4,
                                    T_m=self.pconfig.attention_predictor_length, eps=ln2.eps, want_scores=want_scores)
                    if query_skips > 1:
                        estimated_attention_probs = estimated_attention_probs.repeat_interleave(query_skips, dim=-2)
                        if estimated_attention_score is not None:
                            estimated_attention_score = estimated_attention_score.repeat_interleave(query_skips, dim=-2)
                        t_attention_predictor = t_attention_predictor.repeat_interleave(query_skips, dim=-2)
                bench.register_temp_buffer('t_attention_predictor', t_attention_predictor)
            else:
                with timer("predictor.dec_row"):
                    estimated_attention_score = self.attention_predictor_dec_row(t_attention_predictor)
                    bench.register_temp_buffer('estimated_attention_score_dec_row', estimated_attention_score)
                with timer("predictor.cnn"):
                    estimated_attention_score = self.attention_predictor_cnn(estimated_attention_score)
                    if query_skips > 1:
                        estimated_attention_score = estimated_attention_score.repeat_interleave(query_skips, dim=-2)
                        t_attention_predictor = t_attention_predictor.repeat_interleave(query_skips, dim=-2)
                bench.register_temp_buffer('t_attention_predictor', t_attention_predictor)
        if not use_hip:
            with timer("mask_softmax"):
                estimated_attention_probs = torch.softmax(estimated_attention_score.float(), dim=-1) \
                    .to(estimated_attention_score.dtype)
        bench.register_temp_buffer('estimated_attention_score', estimated_attention_score)
        bench.register_temp_buffer('estimated_attention_probs', estimated_attention_probs)
        return v, t_attention_predictor, estimated_attention_score, estimated_attention_probs

    # ------------------------------------------------------------------------------------------------
    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                q_for_atten: torch.Tensor, k_for_atten: torch.Tensor, v_for_atten: torch.Tensor,
                q_for_score: torch.Tensor, k_for_score: torch.Tensor,
                attention_mask: torch.Tensor, attention_scores_truth: torch.Tensor,
                context_layer_truth: torch.Tensor, last_state=None):
        bench = get_bench()
        dynamic_k = int(os.environ.get('DYNAMIC_K', '0'))
        if dynamic_k > 0:
            warnings.warn(f'dynamic k {dynamic_k}')
            self.pconfig.k = dynamic_k
        if self._warning_messages:
            print(self._warning_messages)
            self._warning_messages = ''
        if not self.pconfig.causal:
            raise NotImplementedError("non-causal (BERT) SEA is outside this build's scope (SURVEY.md 2.1 #12)")
        # the causal model pools the heads of a row ('causal_batch'); the reference turns k_flatten=False into the per-query
        # pooling, which asserts `not causal` (attention.py:786-788, 851), and 'batch' / 'head' do the same (:834, :839)
        assert self.pconfig.k_flatten and self.pconfig.k_flatten_dim == 'causal_batch', \
            f"causal SEA selects per row over the pooled heads (k_flatten=True, k_flatten_dim='causal_batch'); got " \
            f"k_flatten={self.pconfig.k_flatten}, k_flatten_dim={self.pconfig.k_flatten_dim!r}"
        if self.pconfig.use_cache or last_state is not None:
            assert self.pconfig.attention_predictor_method != 'comp', "the compressed predictor has no cached form (attention.py:650)"
            return self._forward_cached(q, k, v, q_for_atten, k_for_atten, v_for_atten, q_for_score, k_for_score,
                                        attention_mask, last_state)

        if context_layer_truth is not None:
            if callable(context_layer_truth):
                with torch.no_grad():
                    context_layer_truth = context_layer_truth()
            context_layer_truth = context_layer_truth.to(q.device, non_blocking=True)
            if callable(attention_scores_truth):
                with torch.no_grad():
                    attention_scores_truth = attention_scores_truth()
            attention_scores_truth = attention_scores_truth.to(q.device, non_blocking=True)

        if q.dtype in (torch.float16, torch.bfloat16):
            FP_MIN = torch.finfo(torch.float16).min / 2        # fp16 floor even for bf16 (:395-396)
        elif q.dtype == torch.float32:
            FP_MIN = torch.finfo(torch.float32).min / 2
        else:
            raise Exception('unknown type')

        N, H1, T_DST, T_SRC = attention_mask.shape
        assert T_DST == T_SRC
        assert H1 == 1
        causal_attention_mask = attention_mask
        attention_mask = attention_mask[:, :, :, :1].transpose(-1, -2)
        dst_attention_mask = causal_attention_mask[:, :, :, :1]
        if self.benchmarking and self.assume_not_padded is not None:
            not_padded = bool(self.assume_not_padded)          # caller knows; skips the reference's sync (:434)
        else:
            not_padded = bool((attention_mask > -1).all().item())

        bench.register_temp_buffer('q', q)
        bench.register_temp_buffer('k', k)
        bench.register_temp_buffer('v', v)
        bench.register_temp_buffer('attention_mask', attention_mask)

        with timer("perlin"):
            N, H, T, HID = q.shape
            v, t_attention_predictor, estimated_attention_score, estimated_attention_probs = self._estimate(
                q, k, v, q_for_atten, k_for_atten, v_for_atten, dst_attention_mask, not_padded, T_SRC)
            T_M = estimated_attention_probs.shape[-1]
            assert estimated_attention_probs.shape[-2] == T_DST

            def resize_dense(x, fill, handle_oversample=True):
                return ops.resize_from_m_to_t(
                    x=x, masked_fill_value=fill, attention_mask=causal_attention_mask, target_width=T_SRC,
                    training=self.training and self.pconfig.causal, is_causal=True, k=self.pconfig.k,
                    oversampled=self.pconfig.k_oversample if handle_oversample else 1.0)

            loss = 0
            estimated_attention_probs_resized = None
            head_chunk = None if self.benchmarking else self.dense_head_chunk
            if not self.benchmarking and attention_scores_truth is not None:
                if head_chunk:
                    # KD loss of the estimator (:741-763) a few heads at a time: both terms are means over (n, h, t[, s]),
                    # so the chunks' values weighted by their share of the heads add up to the unchunked loss
                    for h0 in range(0, H, int(head_chunk)):
                        h1 = min(H, h0 + int(head_chunk))
                        sc = resize_dense(estimated_attention_score[:, h0:h1], FP_MIN, False).float()
                        loss = loss + _kl_and_mse(sc, attention_scores_truth[:, h0:h1], causal_attention_mask, FP_MIN) \
                            * ((h1 - h0) / H)
                        del sc
                else:
                    estimated_attention_probs_resized = resize_dense(estimated_attention_probs, 0, False)
                    estimated_attention_score_resized = resize_dense(estimated_attention_score, FP_MIN, False).float()
                    loss = loss + _kl_and_mse(estimated_attention_score_resized, attention_scores_truth,
                                              causal_attention_mask, FP_MIN)
                    bench.register_temp_buffer('estimated_attention_probs_resized', estimated_attention_probs_resized)
                    bench.register_temp_buffer('estimated_attention_score_resized', estimated_attention_score_resized)

            if not not_padded:
                estimated_attention_probs = estimated_attention_probs * (dst_attention_mask > -1)
            bench.register_temp_buffer('masked_estimated_attention_probs', estimated_attention_probs)

            if self.benchmarking:
                out = self._forward_sparse(q, v, q_for_score, k_for_score, t_attention_predictor,
                                           estimated_attention_probs, dst_attention_mask, not_padded, T_SRC, T_M)
                partial_context_layer, partial_attention_probs, partial_attention_mask = out
                attention_probs_dense = None
            else:
                out = self._forward_dense(q, v, q_for_score, k_for_score, t_attention_predictor,
                                          estimated_attention_probs, causal_attention_mask, dst_attention_mask,
                                          attention_scores_truth, FP_MIN, T_SRC, T_M, resize_dense, head_chunk=head_chunk)
                partial_context_layer, partial_attention_probs, partial_attention_mask, attention_probs_dense, l2 = out
                loss = loss + l2

            bench.register_temp_buffer('partial_context_layer_sparse', partial_context_layer)
            if self.pconfig.random_lookup:
                raise Exception("random_lookup is a dead ablation in the reference as well (`raise Exception()`, :1254-1255)")
            if self.pconfig.context_output_method != 'mix':
                raise Exception("only context_output_method='mix' is live in the reference (:1286-1314)")

            if not self.benchmarking and context_layer_truth is not None:
                loss = loss + F.mse_loss(context_layer_truth, partial_context_layer)

            # (head-chunked dense mode never forms the resized T x T estimate: it hands out the T_M-wide map, like sparse mode)
            estimated_for_output = (estimated_attention_probs if (self.benchmarking or head_chunk)
                                    else estimated_attention_probs_resized)
            bench.register_temp_buffer('partial_context_layer', partial_context_layer)
            assert partial_context_layer.shape[-2] == q.shape[-2]

            return PerlinAttentionOutput(
                loss=loss,
                context_layer=partial_context_layer,
                partial_attention_probs=partial_attention_probs,
                partial_attention_mask=partial_attention_mask,
                estimated_attention_probs_m=estimated_attention_probs,
                estimated_attention_probs=estimated_for_output,
                dense_attention_probs=attention_probs_dense,
                key_for_score=k_for_score,
                state=None,
            )

    # ------------------------------------------------------------------------------------------------
    def decode_session(self, state, key_prefix, value_prefix, capacity: int, use_graph: bool = True):
        """Graph-replayed single-token decoding from `state` (the `.state` of a cached forward over `key_prefix` /
        `value_prefix`, (N,H,L,D)) up to `capacity` tokens: see perlin_attention/decode.py."""
        from .decode import DecodeSession
        return DecodeSession(self, state, key_prefix, value_prefix, capacity, use_graph=use_graph)

    def _decode_keep(self, H, T_DST, T_SRC, T_M):
        """K_t of the new rows only (absolute positions T_SRC-T_DST+1 .. T_SRC), same fp32 expression as
        ops.keep_table_causal (attention.py:849-866), and the capacity bound of their CSR."""
        ctl = torch.arange(T_SRC - T_DST + 1, T_SRC + 1, dtype=torch.long)
        per = H * (self.pconfig.k * self.pconfig.k_oversample * T_M / ctl)
        keep_cpu = torch.clamp_max(torch.clamp_min(torch.round(per), 1), H * T_M).to(torch.int32)
        return keep_cpu, ops.z_capacity(keep_cpu, H, T_DST, T_SRC, T_M, int(self.pconfig.k), True)

    def _forward_cached(self, q, k, v, q_for_atten, k_for_atten, v_for_atten, q_for_score, k_for_score,
                        attention_mask, last_state):
        """KV-cache decoding (reference: the `use_cache` branches of attention.py:391-439,527-572,
        630-646,1229-1235 + attention_state.py).  q / q_for_* bring the T_DST NEW rows, k / v the whole T_SRC prefix;
        `attention_mask` is the (N,1,T_DST,T_SRC) tail of the causal mask.  Returns the rows a stateless forward over
        T_SRC tokens would produce for the last T_DST positions, plus the new state.  The estimator (a handful of
        rows per call) runs on the torch modules; steps H..L run on the HIP kernels with T_dst < T_src."""
        from .attention_state import PerlinAttentionState, PerformerState, CnnWindowState, CumAvgState, cnn_lookback
        if not self.pconfig.causal:
            raise NotImplementedError("kv-cache decoding is defined for the causal configuration only")
        if torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v)):
            raise NotImplementedError("kv-cache decoding is an inference path (no autograd)")
        N, H, T_DST, HID = q.shape
        T_SRC = k.shape[-2]
        assert attention_mask.shape == (N, 1, T_DST, T_SRC) and v.shape[-2] == T_SRC and T_DST <= T_SRC
        state = last_state.clone() if last_state is not None else PerlinAttentionState(self)
        seen = state.seq_len
        assert seen + T_DST == T_SRC, f"state has seen {seen} tokens, call brings {T_DST} new of {T_SRC}"
        bench = get_bench()
        T_M = self.pconfig.attention_predictor_length
        LB = cnn_lookback(self.attention_predictor_cnn)          # 8 rows for two dilated convs, 12 with the deeper hotfix
        ps = state.get(PerlinAttentionState.PERFORMER, PerformerState)
        cs = state.get(PerlinAttentionState.CNN, lambda: CnnWindowState(LB))
        assert cs.lookback >= LB, "state was built for a shallower predictor CNN"
        cav = state.get(PerlinAttentionState.CUMAVG, CumAvgState)
        # 16-bit inference, d = 64: the whole estimator stays on the stateless path's kernels.  The Performer continues
        # its own fp32 sums from a state image (`sea_performer_causal_step`), which also carries the column sums of v
        # for the cumulative average; a call that brings many rows (a prefill) is just a long step.
        if callable(cs.rows_c8):
            assert self._hip_estimator_ok(q), "a window hand-off hook goes with the HIP estimator (16-bit inference)"
        hip_all = (self._hip_estimator_ok(q) and ps.S is None and cs.rows is None and cav.cumsum is None
                   and v_for_atten is v and q_for_atten.dtype == k_for_atten.dtype == v.dtype
                   and ops.performer_avg_supported(q_for_atten, self.performer_nb_features)
                   and self._fused_mlp_ok(torch.empty((N, H, 0, 3 * HID), dtype=q.dtype, device=q.device)))
        if not hip_all:
            assert ps.image is None and cs.rows_c8 is None and not cav.in_image, \
                "this state was written by the HIP estimator (16-bit inference); keep dtype / mode fixed while decoding from it"
        if last_state is None and T_DST > 4 * LB and q.is_cuda and self.benchmarking and not hip_all:
            # ---- prefill: the rows come from the stateless fast path (HIP estimator + kernels, one pass); the state a
            # later call needs is rebuilt from sums over the prefix plus a cached-mode pass over the last LOOKBACK rows
            self.pconfig.use_cache = False
            try:
                out = self.forward(q, k, v, q_for_atten, k_for_atten, v_for_atten, q_for_score, k_for_score,
                                   attention_mask, None, None, None)
            finally:
                self.pconfig.use_cache = True
            with torch.no_grad():
                cut = T_SRC - LB
                pos = self.v_eye_learned_causal[:, :, :cut, :]
                ka, va = k_for_atten[..., :cut, :].float(), v_for_atten[..., :cut, :].float()
                kp = self.performer.feature_map(ka).double()
                ps = PerformerState()
                ps.S = torch.matmul(kp.transpose(-1, -2), torch.cat([pos.expand(va.shape).float(), va], dim=-1).double())
                ps.ksum = kp.sum(-2)
                ps.seq_index = cut
                cav = CumAvgState()
                cav.cumsum, cav.prev_len = v[..., :cut, :].float().sum(-2, keepdim=True), cut
                state.states[PerlinAttentionState.PERFORMER] = ps
                state.states[PerlinAttentionState.CUMAVG] = cav
                tail = self._forward_cached(q[..., cut:, :], k, v, q_for_atten[..., cut:, :], k_for_atten, v_for_atten,
                                            q_for_score[..., cut:, :], k_for_score, attention_mask[:, :, cut:, :], state)
            return PerlinAttentionOutput(*out[:-1], state=tail.state)
        with timer("perlin"), torch.no_grad():
            # ---- A-C: value augmentation + causal Performer on the new rows, running sums carried ------------------
            with timer("performer"):
                sl = slice(T_SRC - T_DST, T_SRC)
                v_new = v[..., sl, :]
                avg_rows = None
                if hip_all:
                    C_ = ops.performer_chunk_rows(HID, self.performer_nb_features, q.dtype)
                    nseg = self.performer_segments or ops.performer_plan(N, H, T_DST + seen % C_, HID, self.performer_nb_features, q.dtype)[0]
                    # chunk-aligned step: the image is the state at the last Performer chunk boundary and the open chunk's
                    # rows are walked again from the kv-cache, so the new rows come out BITWISE as the stateless pass
                    # computes them, whatever the piece sizes (attention_state.py:43-140, test_perlin_opt_cache.py:7-32)
                    performer_value, avg_rows, image = ops.performer_step(
                        q_for_atten, k_for_atten, v, self.v_eye_learned_causal[0, 0],
                        self.performer.projection_matrix, state_in=ps.image, t_base=seen, n_segments=nseg)
                    ps = PerformerState()
                    ps.image, ps.seq_index = image, T_SRC
                else:
                    pos = self.v_eye_learned_causal[:, :, sl, :]
                    qa = q_for_atten.float()
                    ka, va = k_for_atten[..., sl, :].float(), v_for_atten[..., sl, :].float()
                    vaug = torch.cat([pos.expand(va.shape).float(), va], dim=-1)
                    outs = []
                    for c0 in range(0, T_DST, 256):                  # bounded (chunk x chunk) score tiles (prefill)
                        c1 = min(T_DST, c0 + 256)
                        ps, o = ps.step(self.performer.feature_map(qa[..., c0:c1, :]),
                                        self.performer.feature_map(ka[..., c0:c1, :]), vaug[..., c0:c1, :])
                        outs.append(o)
                    performer_context_layer = torch.cat(outs, dim=-2).to(q_for_atten.dtype)
                    performer_value = torch.cat([performer_context_layer, v_new], dim=-1)
                state.states[PerlinAttentionState.PERFORMER] = ps
            # ---- D-G: predictor MLP, windowed CNN, softmax ---------------------------------------------------------
            gates = None
            fused_sel = None
            if hip_all:
                # the stateless path's kernels on the new rows: one-launch MLP (its output is the CNN input AFTER
                # lnorm1, channel-blocked), the two MFMA convolutions over [cached window | new rows] (rows before the
                # window read as zero padding; its LOOKBACK rows cover the convolutions' reach), tail on the new rows
                with timer("predictor"):
                    x, t_attention_predictor, row_scale_, avg_scale_ = ops.predictor_mlp(
                        performer_value, self.attention_predictor_enc[0], self.attention_predictor_enc[1],
                        self.attention_predictor_dec_row[0], self.attention_predictor_cnn[0].module,
                        self.attention_predictor_dec_scaler[0], want_tpred=False)
                    gates = (row_scale_, avg_scale_)
                    # the window may be a hand-off hook instead of a tensor: row-split multi-GPU runs (distributed.py) give
                    # this rank's freshly computed rows to the next rank and receive the previous rank's last rows here
                    win = cs.rows_c8(x) if callable(cs.rows_c8) else cs.rows_c8
                    xs = x if win is None else torch.cat([win, x], dim=1)                     # (N, rows, C/8, W, 8)
                    new_cs = CnnWindowState(cs.lookback)
                    new_cs.rows_c8 = xs[:, -cs.lookback:]
                    state.states[PerlinAttentionState.CNN] = new_cs
                    keepres, ln2 = self.attention_predictor_cnn[1].module, self.attention_predictor_cnn[2].module
                    body = list(keepres.net.children())
                    y = xs
                    for li_ in range(0, len(body) - 2, 2):
                        conv = body[li_].module
                        y = ops.causal_conv_c8(y, conv.weight, conv.bias, conv.kernel_size, conv.dilation,
                                               conv.padding[1], relu=True)
                    conv4 = body[-1].module
                    y_new = y[:, -T_DST:].contiguous()
                    if ops.predictor_tail_select_supported(y_new, H, T_M):
                        # tail + top-k selection of the new rows in one launch (their absolute widths: T_src given)
                        keep_cpu, z_cap = self._decode_keep(H, T_DST, T_SRC, T_M)
                        estimated_attention_probs, _, fused_sel = ops.predictor_tail_select(
                            y_new, conv4.weight[:, :, 0, 0], conv4.bias, ln2.weight, ln2.bias, up=4, T_m=T_M,
                            keep=keep_cpu.to(q.device, non_blocking=True), k=int(self.pconfig.k), T_src=T_SRC,
                            is_causal=True, eps=ln2.eps, want_scores=False)
                    else:
                        estimated_attention_probs, _ = ops.predictor_tail(
                            y_new, conv4.weight[:, :, 0, 0], conv4.bias, ln2.weight, ln2.bias, up=4, T_m=T_M,
                            eps=ln2.eps, want_scores=False)
            else:
                with timer("predictor"):
                    t_attention_predictor = self.attention_predictor_enc(performer_value)
                    x = self.attention_predictor_dec_row(t_attention_predictor)
                    cs, estimated_attention_score = cs.step(self.attention_predictor_cnn, x)
                    state.states[PerlinAttentionState.CNN] = cs
                    estimated_attention_probs = torch.softmax(estimated_attention_score.float(), dim=-1) \
                        .to(estimated_attention_score.dtype).contiguous()
            # ---- H-I: grouped top-k of the new rows (their absolute widths), interpolation to flat CSR -------------
            with timer("interp"):
                if fused_sel is not None:
                    csr = ops.csr_from_selection(*fused_sel, H, T_M, T_SRC, int(self.pconfig.k), True, z_cap,
                                                 defer_emit=self._fuses_interpolation(q, T_M))
                else:
                    keep_cpu, z_cap = self._decode_keep(H, T_DST, T_SRC, T_M)
                    csr, _ = ops.topk_to_csr(estimated_attention_probs, keep_cpu.to(q.device), int(self.pconfig.k),
                                             target_width=T_SRC, is_causal=True, z_cap=z_cap)
            # ---- J-L: gates, cumulative average (carried), fused sparse attention ----------------------------------
            with timer("attention"):
                if gates is not None:
                    row_scale = gates[0] if self.pconfig.partial_attention_scaler else None
                    average_scale = gates[1]
                else:
                    sig = torch.sigmoid(self.attention_predictor_dec_scaler(t_attention_predictor).float())
                    row_scale = sig[..., 0].contiguous() if self.pconfig.partial_attention_scaler else None
                    average_scale = sig[..., 1].contiguous()
                if avg_rows is not None:                          # the Performer launch produced it (sums live in its image)
                    average_context_layer = avg_rows
                    cav = CumAvgState()
                    cav.prev_len, cav.in_image = T_SRC, True
                else:
                    cav, average_context_layer = cav.step(v_new)
                state.states[PerlinAttentionState.CUMAVG] = cav
                qs = q_for_score if q_for_score.stride(-1) == 1 else q_for_score.contiguous()
                ks = k_for_score if k_for_score.stride(-1) == 1 else k_for_score.contiguous()
                vs = v if v.stride(-1) == 1 else v.contiguous()
                ks, vs = ks.to(qs.dtype), vs.to(qs.dtype)
                out_dtype = self.context_layer_dtype or torch.float32
                ctx = torch.empty((N, T_DST, H * HID), dtype=out_dtype, device=q.device)
                want_p = self.return_attention_probs
                plan = None
                if (self.sparse_kernel == "auto" and not want_p and qs.dtype != torch.float32 and HID in (64, 80, 128)
                        and T_DST >= 16 and not csr.col_is_pending):   # same kernel choice as the stateless path
                    plan = ops.attention_plan(csr, T_M, is_causal=True)
                res = ops.sparse_attention(qs, ks, vs, csr, row_scale=row_scale, avg=average_context_layer.to(qs.dtype).contiguous(),
                                           mix=average_scale, out=ctx.view(N, T_DST, H, HID).permute(0, 2, 1, 3),
                                           want_probs=want_p, path="gather" if want_p else self.sparse_kernel, plan=plan)
                probs_csr = csr.with_values(res[1]) if self.return_attention_probs else None
        bench.register_temp_buffer('estimated_attention_probs', estimated_attention_probs)
        mask_out = csr.to_sparse_csr() if self.materialize_csr else csr
        if probs_csr is not None and self.materialize_csr:
            probs_csr = probs_csr.to_sparse_csr()
        return PerlinAttentionOutput(
            loss=0, context_layer=ctx, partial_attention_probs=probs_csr, partial_attention_mask=mask_out,
            estimated_attention_probs_m=estimated_attention_probs, estimated_attention_probs=estimated_attention_probs,
            dense_attention_probs=None, key_for_score=k_for_score, state=state)

    # ------------------------------------------------------------------------------------------------
    def _forward_sparse(self, q, v, q_for_score, k_for_score, t_attention_predictor, probs, dst_attention_mask,
                        not_padded, T_SRC, T_M):
        """Steps H..L on HIP: no host sync, no (N,T,H*T_M) sort, no dense (N,H,T,T) tensor."""
        bench = get_bench()
        N, H, T, HID = q.shape
        probing = bench.activate_temp_buffers
        with timer("mask"):
            keep, z_cap = self._keep_table(H, T, T_SRC, T_M, q.device)
            bench.register_temp_buffer('per_item_top_k', None, lazy=lambda: keep.float().view(1, T, 1))
        with timer("interp"):
            if probs.stride(-1) != 1:
                probs = probs.contiguous()
            if not not_padded:
                # padded query rows keep nothing (the reference zeroes those mask rows, :927-931)
                keep = (keep.view(1, T) * (dst_attention_mask.view(N, T) > -1).to(torch.int32)).contiguous()
            fs = self._fused_selection
            if fs is not None and fs[0] is probs and not_padded and not probing:
                # the selection already ran inside the predictor-tail launch on this very map: scan + emit only
                # `sparse_kernel = "gather"`: the column array is left to the attention launch, whose gather kernels do the
                # interpolation of their own (row, head) themselves and write `col` (sea_sparse_attention_fused: steps I + J
                # in one launch); with any other consumer the handle runs the emit launch on first use of `.col`
                csr, mask_m = ops.csr_from_selection(*fs[1], H, T_M, T_SRC, int(self.pconfig.k), True, z_cap,
                                                     defer_emit=self._fuses_interpolation(q, T_M)), None
            else:
                # (the stand-alone selection: fp32 data, padded batches, QUERY_SKIPS, probing.  Its columns, too, are left to
                # the fused attention launch wherever that form exists -- round 5: fp32 data paid a 73 us emit launch for a
                # column array the fused launch then did not even read)
                csr, mask_m = ops.topk_to_csr(probs, keep, int(self.pconfig.k), target_width=T_SRC, is_causal=True,
                                              z_cap=z_cap, want_mask=probing,
                                              defer_emit=self._fuses_interpolation(q, T_M) and not probing)
            self._fused_selection = None
        if probing:
            bench.register_temp_buffer('partial_attention_mask_before_interp', mask_m)
            bench.register_temp_buffer('partial_attention_mask', None,
                                       lazy=lambda: ops.flat_csr_to_dense(csr, T_SRC, H))
        bench.register_temp_buffer('q_for_score', q_for_score)
        bench.register_temp_buffer('k_for_score', k_for_score)

        with timer("attention"):
            with timer('attention.sparse.scaler'):
                if self._fused_gates is not None:                       # already produced by the fused predictor MLP
                    row_scale, average_scale = self._fused_gates
                    if not self.pconfig.partial_attention_scaler:
                        row_scale = None
                    # probing (temp buffers on) keeps the pre-sigmoid tensor of the reference available
                    estimated_scales = (self.attention_predictor_dec_scaler(t_attention_predictor)
                                        if t_attention_predictor is not None else None)
                else:
                    estimated_scales = self.attention_predictor_dec_scaler(t_attention_predictor)      # (N,H,T,2)
                    sig = torch.sigmoid(estimated_scales.float())
                    row_scale = sig[..., 0].contiguous() if self.pconfig.partial_attention_scaler else None
                    average_scale = sig[..., 1].contiguous()
            with timer("attention.avg_pool"):
                if self._avg_ahead is not None and not_padded:         # came out of the Performer launch
                    average_context_layer, self._avg_ahead = self._avg_ahead, None
                else:
                    avg_v = v if not_padded else v * (dst_attention_mask > -1)
                    if torch.is_grad_enabled() and avg_v.requires_grad:   # training: the average carries gradient to v
                        average_context_layer = (avg_v.float().cumsum(-2) / torch.arange(
                            1, T + 1, device=v.device, dtype=torch.float32).view(1, 1, -1, 1)).to(v.dtype)
                    else:
                        average_context_layer = ops.cumavg(avg_v)      # HIP scan, fp32 accumulation
            out_dtype = self.context_layer_dtype or torch.float32
            qs = q_for_score if q_for_score.stride(-1) == 1 else q_for_score.contiguous()
            ks = k_for_score if k_for_score.stride(-1) == 1 else k_for_score.contiguous()
            vs = v if v.stride(-1) == 1 else v.contiguous()
            if ks.dtype != qs.dtype:
                ks = ks.to(qs.dtype)
            if vs.dtype != qs.dtype:
                vs = vs.to(qs.dtype)
            if average_context_layer.dtype != qs.dtype:
                average_context_layer = average_context_layer.to(qs.dtype)
            want_probs = self.return_attention_probs or probing
            probs_csr = None
            with timer('attention.sparse.fused'):
                if probing:
                    p1, pvals = ops.sparse_attention(qs, ks, vs, csr, row_scale=row_scale, want_probs=True)   # fp32 (N,H,T,D)
                    probs_csr = csr.with_values(pvals)
                    bench.register_temp_buffer('partial_context_layer_1', p1)
                    a = average_scale.unsqueeze(-1)
                    p2 = p1 * a + (1 - a) * average_context_layer
                    bench.register_temp_buffer('estimated_scales', estimated_scales)
                    bench.register_temp_buffer('average_scale', a)
                    bench.register_temp_buffer('average_context_layer', average_context_layer)
                    bench.register_temp_buffer('partial_context_layer_2', p2)
                    with timer("context_permute"):
                        ctx = p2.permute(0, 2, 1, 3).contiguous().view(N, T, H * HID).to(out_dtype)
                elif torch.is_grad_enabled() and any(t_.requires_grad for t_ in (qs, ks, vs, average_scale, average_context_layer)
                                                    + ((row_scale,) if row_scale is not None else ())):
                    # training through the sparse branch: HIP forward + backward for the sparse product
                    # (sea_sparse_attention_bwd), row scale and mix as torch ops so that autograd owns their gradients
                    p2 = ops.sparse_attention_autograd(qs, ks, vs, csr, row_scale=row_scale, avg=average_context_layer,
                                                       mix=average_scale)
                    ctx = p2.permute(0, 2, 1, 3).reshape(N, T, H * HID).to(out_dtype)
                else:
                    # write straight into the (N, T, H*D) layout of :1279-1282
                    ctx = torch.empty((N, T, H * HID), dtype=out_dtype, device=q.device)
                    plan = None
                    if (self.sparse_kernel == "auto" and not want_probs and qs.dtype != torch.float32
                            and HID in (64, 80, 128) and not csr.col_is_pending):
                        # kernel choice on the device: a small launch over the selection's pixel masks counts the 16-row
                        # blocks whose rows share most of their keys; the MFMA tile kernel runs the launch when they are
                        # the majority, the gather kernels otherwise (the idle kernel's workgroups exit at once)
                        plan = ops.attention_plan(csr, T_M, is_causal=True)
                    res = ops.sparse_attention(qs, ks, vs, csr, row_scale=row_scale, avg=average_context_layer,
                                               mix=average_scale, out=ctx.view(N, T, H, HID).permute(0, 2, 1, 3),
                                               want_probs=want_probs, path="gather" if want_probs else self.sparse_kernel,
                                               plan=plan, keep_columns_pending=self.lazy_csr_columns and not want_probs)
                    if want_probs:
                        probs_csr = csr.with_values(res[1])
        if probs_csr is not None:
            bench.register_temp_buffer('partial_attention_probs', None, lazy=lambda: ops.flat_csr_to_dense(probs_csr, T_SRC, H))
        if self.materialize_csr:
            return ctx, (probs_csr.to_sparse_csr() if probs_csr is not None else None), csr.to_sparse_csr()
        return ctx, probs_csr, csr

    # ------------------------------------------------------------------------------------------------
    def _forward_dense(self, q, v, q_for_score, k_for_score, t_attention_predictor, probs, causal_attention_mask,
                       dst_attention_mask, attention_scores_truth, FP_MIN, T_SRC, T_M, resize_dense, head_chunk=None):
        """The reference's dense branch (:774-947 sort-based top-k, :948-959 gather resize, :1061-1133).
        `head_chunk`: the T x T part (mask resize, scores, both softmaxes, P.V, KD loss) runs that many heads at a time --
        same arithmetic per head, bounded memory; `partial_attention_probs`, `partial_attention_mask` and
        `dense_attention_probs` (the (N,H,T,T) outputs) are then None."""
        bench = get_bench()
        N, H, T, HID = q.shape
        k_, os_ = self.pconfig.k, self.pconfig.k_oversample
        loss = 0
        with timer("mask"):
            t = probs.transpose(1, 2).reshape(N, T, H * T_M)
            ctl = torch.arange(1, T + 1, dtype=torch.long, device=q.device).view(1, T, 1)
            per_item_top_k = torch.clamp_min(torch.round(H * (k_ * os_ * T_M / ctl)), 1)
            bench.register_temp_buffer('per_item_top_k', per_item_top_k)
            with timer("mask.topk"):
                _, indices = torch.sort(t.float(), dim=-1, descending=True, stable=True)
            rank = torch.empty_like(indices)
            rank.scatter_(-1, indices, torch.arange(H * T_M, device=q.device).view(1, 1, -1).expand_as(indices))
            t_dead_mask = rank >= per_item_top_k
            bench.register_temp_buffer('t_dead_mask', None, lambda: t_dead_mask.float())
            partial_attention_mask = (t_dead_mask.to(q.dtype) * FP_MIN).view(N, T, H, T_M).transpose(1, 2)
            partial_attention_mask = partial_attention_mask.masked_fill(dst_attention_mask < -1, FP_MIN)
        bench.register_temp_buffer('partial_attention_mask_before_interp', partial_attention_mask)
        if head_chunk:
            return self._dense_attention_by_head_chunks(int(head_chunk), v, q_for_score, k_for_score, t_attention_predictor,
                                                        partial_attention_mask, causal_attention_mask, dst_attention_mask,
                                                        attention_scores_truth, FP_MIN, resize_dense)
        with timer("interp"):
            partial_attention_mask = resize_dense(partial_attention_mask, FP_MIN, True)
            partial_attention_mask = partial_attention_mask.masked_fill(causal_attention_mask < -1, FP_MIN)
        bench.register_temp_buffer('partial_attention_mask', partial_attention_mask)
        bench.register_temp_buffer('q_for_score', q_for_score)
        bench.register_temp_buffer('k_for_score', k_for_score)
        with timer("attention"):
            attention_scores_dense = torch.matmul(q_for_score, k_for_score.transpose(-1, -2))
            if attention_scores_truth is not None:
                loss = loss + _kl_and_mse(attention_scores_dense, attention_scores_truth, causal_attention_mask, FP_MIN)
            bench.register_temp_buffer('attention_scores_dense', attention_scores_dense)
            attention_probs_dense = torch.softmax((attention_scores_dense + causal_attention_mask).float(), -1) \
                .to(attention_scores_dense.dtype)
            partial_attention_scores = attention_scores_dense + partial_attention_mask
            partial_attention_probs = torch.softmax(partial_attention_scores.float(), -1).to(partial_attention_scores.dtype)
            partial_attention_probs = partial_attention_probs.masked_fill(partial_attention_mask < -1, 0)
            bench.register_temp_buffer('partial_attention_scores', partial_attention_scores)
            bench.register_temp_buffer('attention_matrix', partial_attention_probs)
            estimated_scales = self.attention_predictor_dec_scaler(t_attention_predictor)
            if self.pconfig.partial_attention_scaler:
                partial_attention_probs = partial_attention_probs * torch.sigmoid(estimated_scales[..., 0:1])
            partial_context_layer = torch.matmul(partial_attention_probs, v)
            bench.register_temp_buffer('partial_context_layer_1', partial_context_layer)
            with timer("attention.avg_pool"):
                avg_v = v * (dst_attention_mask > -1)
                average_context_layer = (avg_v.cumsum(-2) / torch.arange(1, T + 1, device=v.device).view(1, 1, -1, 1)).to(v.dtype)
                average_scale = torch.sigmoid(estimated_scales[..., 1:2])
                partial_context_layer = partial_context_layer * average_scale + (1 - average_scale) * average_context_layer
                bench.register_temp_buffer('estimated_scales', estimated_scales)
                bench.register_temp_buffer('average_scale', average_scale)
                bench.register_temp_buffer('average_context_layer', average_context_layer)
                bench.register_temp_buffer('partial_context_layer_2', partial_context_layer)
        with timer("context_permute"):
            partial_context_layer = partial_context_layer.permute(0, 2, 1, 3).contiguous().view(N, T, H * HID)
        return partial_context_layer, partial_attention_probs, partial_attention_mask, attention_probs_dense, loss

    def _dense_attention_by_head_chunks(self, chunk, v, q_for_score, k_for_score, t_attention_predictor, mask_m,
                                        causal_attention_mask, dst_attention_mask, attention_scores_truth, FP_MIN,
                                        resize_dense):
        """Steps I..L of the dense branch (:948-959, :1061-1133, :1236-1244) `chunk` heads at a time.  The top-k mask `mask_m`
        (N,H,T,T_M) pooled over ALL heads is already there (it lives on the T_M-wide map); everything T x T exists for one
        chunk only.  Returns the tuple of `_forward_dense` with None for the three T x T outputs."""
        bench = get_bench()
        N, H, T, HID = v.shape[0], v.shape[1], q_for_score.shape[2], v.shape[3]
        loss = 0
        estimated_scales = self.attention_predictor_dec_scaler(t_attention_predictor)
        bench.register_temp_buffer('q_for_score', q_for_score)
        bench.register_temp_buffer('k_for_score', k_for_score)
        parts = []
        with timer("attention"):
            for h0 in range(0, H, chunk):
                h1 = min(H, h0 + chunk)
                with timer("interp"):
                    pm = resize_dense(mask_m[:, h0:h1], FP_MIN, True)
                    pm = pm.masked_fill(causal_attention_mask < -1, FP_MIN)
                scores = torch.matmul(q_for_score[:, h0:h1], k_for_score[:, h0:h1].transpose(-1, -2))
                if attention_scores_truth is not None:
                    loss = loss + _kl_and_mse(scores, attention_scores_truth[:, h0:h1], causal_attention_mask, FP_MIN) \
                        * ((h1 - h0) / H)
                ps = scores + pm
                pp = torch.softmax(ps.float(), -1).to(ps.dtype)
                pp = pp.masked_fill(pm < -1, 0)
                del ps, pm, scores
                if self.pconfig.partial_attention_scaler:
                    pp = pp * torch.sigmoid(estimated_scales[:, h0:h1, :, 0:1])
                parts.append(torch.matmul(pp, v[:, h0:h1]))
                del pp
            partial_context_layer = torch.cat(parts, dim=1)
            bench.register_temp_buffer('partial_context_layer_1', partial_context_layer)
            with timer("attention.avg_pool"):
                avg_v = v * (dst_attention_mask > -1)
                average_context_layer = (avg_v.cumsum(-2) / torch.arange(1, T + 1, device=v.device).view(1, 1, -1, 1)).to(v.dtype)
                average_scale = torch.sigmoid(estimated_scales[..., 1:2])
                partial_context_layer = partial_context_layer * average_scale + (1 - average_scale) * average_context_layer
                bench.register_temp_buffer('estimated_scales', estimated_scales)
                bench.register_temp_buffer('average_scale', average_scale)
                bench.register_temp_buffer('average_context_layer', average_context_layer)
                bench.register_temp_buffer('partial_context_layer_2', partial_context_layer)
        with timer("context_permute"):
            partial_context_layer = partial_context_layer.permute(0, 2, 1, 3).contiguous().view(N, T, H * HID)
        return partial_context_layer, None, None, None, loss
