"""Single-token decoding of one SEA attention layer as a replayed HIP graph.

The reference generates one position per forward (src/main/opt_generate.py:131 -> the `use_cache` branches of
perlin_attention/attention.py + attention_state.py).  `PerlinAttention._forward_cached` is that path on the HIP
kernels: a dozen launches per position whose arguments change with the position (T_src, the value-embedding row, K_t,
tensor shapes that grow), so the step is bound by the host enqueueing them.

`DecodeSession` freezes everything position-dependent into DEVICE memory instead:

  * K / V caches of a fixed capacity; the new row is written at a device-resident index,
  * the Performer state image and the predictor CNN's window are updated in place,
  * the three kernels that need the position read it from a device int32
    (`sea_performer_causal_step_at`, `sea_predictor_tail_select_at`, `sea_csr_emit_at`, include/sea_hip.h),
  * column ids are encoded against the cache capacity, so the fused attention launch has static arguments.

The whole step is then captured ONCE with `torch.cuda.graph` and replayed per token.  Round 4: the framework glue around
the kernels (three input copies, index_copy_, cat + copy of the CNN window, row scan, two counter adds: eleven ~4.5 us
launches, 50 of a position's 130 us) is two launches now -- `sea_decode_stage` in front of the replay (the only launch whose
arguments change: it takes the caller's q row and appends k / v to the caches) and `sea_c8_window_shift` at its end (it
also advances the two device counters); the MLP writes the new CNN row straight behind the window and the tail + selection
launch writes the one-row `crow` itself.  Results are bitwise those of `_forward_cached` called position by position
(tests/test_decode_session.py).

Round 5: conv1, conv2, tail + selection and the window shift -- four launches, 37 of a position's 95 us at batch 1, all of it
fixed cost (each convolution staged its 74 KB weight image into LDS to convolve a 25-row window of which one row is new) --
are ONE launch, `sea_decode_cnn_tail_select`: a workgroup per sequence computes the one new row of each convolution (the
earlier rows it needs, t - 2 and t - 4, live in two small rings: the MLP's previous outputs and conv1's previous outputs),
runs the unchanged tail + selection on it and advances the device counters.  Nothing is shifted any more: a ring slot is
position % ring size.  Still bitwise (`csrc/sea_convfrag.hpp: conv_row_c8` reproduces the convolution kernel's operand
placement and k order).  Shapes outside that kernel (three-convolution bodies, H > 40) keep the round-4 launches.
Round 5, later: the attention launch of a position is `sea_sparse_attention_fused_at` -- the gather kernel expands the one
row's kept pixels itself (row widths from the device counter), its idle lane groups warm the K / V rows of the expanded lists
-- so the CSR row's column ids are not on the critical path any more (no emit phase / launch; `session.csr.col` emits on
first read).  `fused_attention=False` keeps the emit + unfused launch pair (bitwise the same context).
"""
from typing import Optional

import torch

from . import ops
from .attention_state import PerlinAttentionState, cnn_lookback


class DecodeSession:
    """Built from the state of a cached forward (`PerlinAttentionOutput.state`, HIP estimator: 16-bit inference) and the
    K / V prefix that forward saw.  `step(q, k, v)` takes the NEW row of each tensor, (N, H, 1, D), and returns the
    context row (N, 1, H*D) -- a static buffer, overwritten by the next step."""

    def __init__(self, attention, state: PerlinAttentionState, key_prefix: torch.Tensor, value_prefix: torch.Tensor,
                 capacity: int, use_graph: bool = True, fused_attention: bool = True):
        at = self.attention = attention
        pc = at.pconfig
        assert pc.causal and not at.training, "decoding is the causal inference path"
        N, H, L, D = key_prefix.shape
        assert value_prefix.shape == key_prefix.shape and key_prefix.is_cuda
        assert state is not None and state.seq_len == L, "the state must have seen exactly the prefix"
        LB = cnn_lookback(at.attention_predictor_cnn)
        ps = state.states.get(PerlinAttentionState.PERFORMER)
        cs = state.states.get(PerlinAttentionState.CNN)
        assert ps is not None and ps.image is not None and cs is not None and torch.is_tensor(cs.rows_c8), \
            "the session continues a state written by the HIP estimator (16-bit inference, supported head size)"
        assert cs.rows_c8.shape[1] == LB, f"the prefix must be at least the predictor CNN's reach ({LB} rows)"
        assert L < capacity <= at.v_eye_learned_causal.shape[2], "capacity: beyond the prefix, within the value embedding"
        self.N, self.H, self.D, self.capacity = N, H, D, int(capacity)
        self.T_M = int(pc.attention_predictor_length)
        self.k = int(pc.k)
        dev, dt = key_prefix.device, key_prefix.dtype
        assert dt in (torch.float16, torch.bfloat16)
        assert ops.predictor_tail_select_supported(cs.rows_c8, H, self.T_M, decode=True), "fused tail + selection shape (T_M = 256, H <= 64)"
        self.image = ps.image.clone()                                        # Performer sums, updated in place
        # CNN input rows: the window (last LB rows) and, behind it, the row of the current position -- ONE buffer, so that the
        # MLP writes the new row in place (no cat) and the window moves by an in-place shift at the end of the step
        self.LB = LB
        # round 5: the fused CNN + tail + selection launch (module docstring).  x ring: the MLP's rows of the last LB positions,
        # row of position p in slot p % LB (what `win` holds, by position instead of by age); y1 ring: conv1's rows of the last
        # positions (8 slots: t - 2 dil and t - 4 dil... t must sit in distinct slots for dilation 2)
        body = list(at.attention_predictor_cnn[1].module.net.children())
        convs = [body[i].module for i in range(0, len(body) - 2, 2)]
        C = cs.rows_c8.shape[2] * 8
        self.fused_cnn = (len(convs) == 2 and all(c.kernel_size == 3 and c.in_channels == C and c.out_channels == C
                                                  and isinstance(c.dilation, int) and c.dilation == convs[0].dilation
                                                  and c.padding[1] == c.dilation for c in convs)
                          and ops.decode_cnn_supported(C, H, self.T_M, dt) and LB > 2 * convs[0].dilation)
        if self.fused_cnn:
            dil, RY = convs[0].dilation, 2 * 2 * convs[0].dilation + 1      # t, t - dil, t - 2 dil in distinct slots: 9 for dil 2
            row_shape = tuple(cs.rows_c8.shape[2:])
            pos = torch.arange(L - LB, L, device=dev)
            self.x_ring = torch.zeros((N, LB) + row_shape, dtype=dt, device=dev)
            self.x_ring[:, pos % LB] = cs.rows_c8                            # window row i is position L - LB + i
            y1 = ops.causal_conv_c8(cs.rows_c8.contiguous(), convs[0].weight, convs[0].bias, 3, dil, dil, relu=True)
            self.y1_ring = torch.zeros((N, RY) + row_shape, dtype=dt, device=dev)
            keep_rows = min(RY - 1, LB - 2 * dil)                            # rows whose taps lie inside the window: conv1's true values
            p1 = torch.arange(L - keep_rows, L, device=dev)
            self.y1_ring[:, p1 % RY] = y1[:, LB - keep_rows:]
            self.x_new = torch.zeros((N, 1) + row_shape, dtype=dt, device=dev)   # where the MLP writes the new row
            self.y2 = torch.zeros((N,) + row_shape, dtype=dt, device=dev)
            self.ticket = torch.zeros((1,), dtype=torch.int32, device=dev)
            self.xs = None
        else:                                                      # the round-4 launches: a window the step shifts by one row
            self.xs = torch.zeros((N, LB + 1) + tuple(cs.rows_c8.shape[2:]), dtype=dt, device=dev)
            self.xs[:, :LB] = cs.rows_c8
        # K and V caches are the two halves of ONE tensor and the two position counters two elements of one
        self.kv_cache = torch.zeros((2, N, H, capacity, D), dtype=dt, device=dev)
        self.k_cache, self.v_cache = self.kv_cache[0], self.kv_cache[1]
        self.k_cache[:, :, :L] = key_prefix
        self.v_cache[:, :, :L] = value_prefix
        # device counters: seen = rows the state has seen = cache row of the new token, tsrc = keys the new row sees; the
        # LAST launch of a step (the window shift) advances both
        # (third element, fused CNN launch only: T_src of the step that launch has just closed -- what the emit behind it reads)
        self.ctr32 = torch.tensor([L, L + 1, L + 1], dtype=torch.int32, device=dev)
        self.seen32 = self.ctr32[0:1]
        self.tsrc32 = self.ctr32[1:2]
        self.tsrc_done32 = self.ctr32[2:3]
        self.crow = torch.zeros((N, 2), dtype=torch.int32, device=dev)        # one-row CSR: [0, row total], written by the selection
        self.length = L                                                      # host mirror (bounds check only)
        # K_t of every reachable position (attention.py:849-866, the same fp32 expression as the stateless path) and
        # the largest CSR row any of them can emit
        keep_cpu, _ = at._decode_keep(H, capacity, capacity, self.T_M)
        self.keep_table = keep_cpu.to(dev)
        w = torch.arange(1, capacity + 1)
        per_pixel = torch.clamp_max(torch.div(w + self.T_M - 1, self.T_M, rounding_mode="floor"), self.k)
        bound = torch.minimum(keep_cpu.to(torch.long) * per_pixel, H * torch.minimum(w, torch.tensor(self.T_M * self.k)))
        self.z_cap = max(int(bound.max().item()), 1)
        self.q_in = torch.zeros((N, H, 1, D), dtype=dt, device=dev)
        self.ctx = torch.zeros((N, 1, H * D), dtype=at.context_layer_dtype or torch.float32, device=dev)
        self.fused_attention = bool(fused_attention)   # the attention launch's decode form (False: emit + the unfused launch)
        self.csr = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.probs = None                                                    # estimated attention probabilities of the last step
        self._pinned, self._prep_generation = None, ops.prep_generation()
        if use_graph:
            self._capture()

    @property
    def win(self) -> torch.Tensor:
        """The predictor CNN's input rows of the last LB positions, oldest first, (N, LB, C/8, W, 8): a view of the shifted
        window (round-4 launches) or the ring read out by age (fused CNN launch; a copy)."""
        if self.fused_cnn:
            pos = torch.arange(self.length - self.LB, self.length, device=self.x_ring.device)
            return self.x_ring[:, pos % self.LB]
        return self.xs[:, :self.LB]

    # the one launch of a position whose arguments change: q -> q_in, k / v -> the caches' new row
    def _stage(self, q, k, v):
        ops.decode_stage(q, k, v, self.q_in, self.kv_cache, self.ctr32[:2])

    # the (captured) launches of one position; everything position-dependent is read from device memory
    def _launch(self):
        at, H, D, T_M = self.attention, self.H, self.D, self.T_M
        # chunk-aligned step: the kernel walks the open Performer chunk again from the caches (which hold the new row already)
        performer_value, avg_rows, _ = ops.performer_step(
            self.q_in, self.k_cache, self.v_cache, at.v_eye_learned_causal[0, 0], at.performer.projection_matrix,
            state_in=self.image, t_base_dev=self.seen32)
        _x, _t, row_scale, avg_scale = ops.predictor_mlp(
            performer_value, at.attention_predictor_enc[0], at.attention_predictor_enc[1],
            at.attention_predictor_dec_row[0], at.attention_predictor_cnn[0].module,
            at.attention_predictor_dec_scaler[0], want_tpred=False,
            x_c8_out=self.x_new if self.fused_cnn else self.xs[:, -1:])       # the new row (fused CNN: its own buffer; else behind the window)
        keepres, ln2 = at.attention_predictor_cnn[1].module, at.attention_predictor_cnn[2].module
        body = list(keepres.net.children())
        conv4 = body[-1].module
        if self.fused_cnn:
            # conv1 + conv2 (one new row each) + tail + selection + the counters' advance: one launch
            # The attention launch expands the kept pixels itself (sea_sparse_attention_fused_at: the emit phase / launch and the
            # crow -> col -> K / V chain leave the position's critical path); where that form does not exist the CSR row's
            # column ids come out of this launch (LDS allowing) or an emit launch behind it, as in the first round-5 version.
            fused_attn = self.fused_attention and ops.fused_interp_supported(self.q_in.dtype, D, T_M)
            emits = not fused_attn and ops.decode_cnn_emits(self.x_new.shape[-3] * 8)
            col = torch.empty((self.N, self.z_cap), dtype=torch.int32, device=self.q_in.device) if emits else None
            self.probs, sel = ops.decode_cnn_tail_select(
                self.x_new, self.x_ring, self.y1_ring, self.y2, body[0].module, body[2].module, conv4.weight[:, :, 0, 0], conv4.bias,
                ln2.weight, ln2.bias, T_M, self.keep_table, self.k, self.ctr32, self.ticket, self.crow, eps=ln2.eps,
                col_out=col, T_cap=self.capacity)
            if emits:
                csr = ops.FlatCSR(self.crow, col, sel[2], H, self.capacity, bits=sel[0], row_nnz=sel[1])
            else:
                csr = ops.csr_from_selection(*sel, H, T_M, self.capacity, self.k, True, self.z_cap, t_src_dev=self.tsrc_done32,
                                             crow=self.crow, defer_emit=fused_attn)
            ops.sparse_attention(self.q_in, self.k_cache, self.v_cache, csr,
                                 row_scale=row_scale if at.pconfig.partial_attention_scaler else None,
                                 avg=avg_rows, mix=avg_scale, out=self.ctx.view(self.N, 1, H, D).permute(0, 2, 1, 3),
                                 path="gather", keep_columns_pending=True)
            self.csr = csr                                                    # (the step's selection: columns on first read of .col)
            return
        y = self.xs                                                           # (N, LB + 1, C/8, W, 8)
        for i in range(0, len(body) - 2, 2):
            conv = body[i].module
            y = ops.causal_conv_c8(y, conv.weight, conv.bias, conv.kernel_size, conv.dilation, conv.padding[1], relu=True)
        y_new = y[:, -1:]                                                     # (the tail reads the row where it lies)
        self.probs, _, sel = ops.predictor_tail_select(
            y_new, conv4.weight[:, :, 0, 0], conv4.bias, ln2.weight, ln2.bias, up=4, T_m=T_M, keep=self.keep_table,
            k=self.k, T_src=0, is_causal=True, eps=ln2.eps, want_scores=False, t_src_dev=self.tsrc32, crow_out=self.crow)
        csr = ops.csr_from_selection(*sel, H, T_M, self.capacity, self.k, True, self.z_cap, t_src_dev=self.tsrc32, crow=self.crow)
        ops.sparse_attention(self.q_in, self.k_cache, self.v_cache, csr,
                             row_scale=row_scale if at.pconfig.partial_attention_scaler else None,
                             avg=avg_rows, mix=avg_scale, out=self.ctx.view(self.N, 1, H, D).permute(0, 2, 1, 3),
                             path="gather")
        ops.c8_window_shift(self.xs, counters=self.ctr32[:2])                 # the window of the next position; counters += 1

    def _capture(self):
        """One eager step on a side stream would advance the state, so the capture runs against SAVED copies of the
        mutable buffers, restored afterwards (a capture records launches, it does not execute them).

        The captured launches hold RAW POINTERS to the re-laid-out predictor weights of `ops.predictor._prep_cache` (MLP,
        convolution, tail, LayerNorm and projection packs built lazily inside `_launch`).  The session pins the very
        tensors its launches took from the cache (`ops.pinned_prep`), so a cache eviction -- `clear_prep_cache()` from
        another layer's `.to()` / `load_state_dict`, or the cache's own size bound -- cannot free memory a replay still
        reads; and it remembers the cache generation: `step()` re-captures when that has moved, because a cleared cache
        means the weights may have been edited and the pinned packs may be stale."""
        mutable = [self.image, self.kv_cache, self.ctr32]
        mutable += [self.x_ring, self.y1_ring, self.ticket] if self.fused_cnn else [self.xs]
        saved = [t.clone() for t in mutable]
        with ops.pinned_prep() as pins:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():                    # warm-up: lazy library work happens outside the capture
                zero = torch.zeros_like(self.q_in)
                self._stage(zero, zero, zero)
                self._launch()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                self._launch()
        for dst, src in zip(mutable, saved):
            dst.copy_(src)
        self.graph = g
        self.captures = getattr(self, "captures", 0) + 1
        self._pinned = pins
        self._prep_generation = ops.prep_generation()

    def export_state(self) -> PerlinAttentionState:
        """The session's state as the `PerlinAttentionState` a cached forward continues from (copies: the session keeps
        running on its own buffers)."""
        from .attention_state import PerformerState, CnnWindowState, CumAvgState
        st = PerlinAttentionState(self.attention)
        ps = PerformerState()
        ps.image, ps.seq_index = self.image.clone(), self.length
        cs = CnnWindowState(self.LB)
        cs.rows_c8 = self.win.clone()
        cav = CumAvgState()
        cav.prev_len, cav.in_image = self.length, True
        st.states = {PerlinAttentionState.PERFORMER: ps, PerlinAttentionState.CNN: cs, PerlinAttentionState.CUMAVG: cav}
        return st

    @torch.no_grad()
    def step(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        assert self.length < self.capacity, "cache capacity reached"
        if self.graph is not None and ops.prep_generation() != self._prep_generation:
            self.graph = None                                      # (re-captured below, BEFORE this step's stage launch: the capture
            self._capture()                                        #  runs a warm-up step on saved copies of the state)
        self._stage(q, k, v)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._launch()
        self.length += 1
        return self.ctx


class SessionState:
    """What a graph-replayed step hands back in the place of a `PerlinAttentionState` (e.g. as the third element of the
    OPT block's cache tuple): a ticket for the NEXT step of the same session.  It is valid while the session has not moved
    on; `materialize()` turns it into a real state for a call the session cannot serve (several tokens at once)."""

    def __init__(self, session: DecodeSession):
        self.session = session
        self.seq_len = session.length

    @property
    def current(self) -> bool:
        return self.session.length == self.seq_len

    def materialize(self) -> PerlinAttentionState:
        assert self.current, "this decode state is stale: its session has produced later positions (sessions cannot branch)"
        return self.session.export_state()
