"""Flat-CSR operators of SEA on MI355X -- host side of libsea_hip.so.

Mirrors the reference's operator package one to one (same names, argument meaning, assertions):
    src/models/perlin_attention/ops/__init__.py:1-7
and adds the fused entry points the attention module uses (`topk_to_csr`, `sparse_attention`).

Wire format (reference: ops/kernels/causal_resize_m_to_t.py:757-762): a batched
`torch.sparse_csr_tensor` of logical shape (N, T_dst, H*T_src), int64 indices, column id =
head*T_src + key.  Internally the kernels use the int32 `FlatCSR` below and only widen to int64
when a torch CSR tensor is asked for.

Every operator here launches HIP kernels through the C ABI; CPU tensors are rejected.
"""
import math
from ctypes import c_void_p, c_int64
from typing import Optional

import torch

from ... import _lib


def _p(t: Optional[torch.Tensor]):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


# ------------------------------------------------------------------------------------------------
# internal format
# ------------------------------------------------------------------------------------------------
class FlatCSR:
    """int32 flat CSR + per-(row, head) offsets, as produced by the fused top-k/interpolate path.

    crow     (N, T_dst+1) int32   row starts inside `col[n]`
    col      (N, z_cap)   int32   head*T_src + key; entries >= crow[n,-1] are undefined
    head_off (N, T_dst, H+1) int32
    vals     (N, z_cap)   fp32 or None   per-entry values (None = ones): `partial_attention_probs` carries the
                                         rs * softmax values here (attention.py:1162-1171)

    Code written against the reference's `torch.sparse_csr_tensor` reads `.crow_indices() / .col_indices() / .values()`,
    `.shape`, `.is_sparse_csr`: those work on this handle too (int64, trimmed to Z = max nnz and zero padded exactly like
    the wire format; the first such call costs the one host sync the reference pays at causal_resize_m_to_t.py:667)."""

    def __init__(self, crow, col, head_off, H, T_src, bits=None, row_nnz=None, vals=None):
        self.crow, self._col, self.head_off = crow, col, head_off
        self.H, self.T_src = H, T_src
        self.bits, self.row_nnz = bits, row_nnz
        self.vals = vals
        self.N, self.T_dst = crow.shape[0], crow.shape[1] - 1
        self._wire = None
        # (T_m, max_k, is_causal, emit launcher) while the column array has not been written yet: `csr_from_selection(...,
        # defer_emit=True)` hands such a handle to the fused attention launch, which writes `col` itself; whoever reads
        # `.col` first otherwise runs the emit launch then (same stream: ordered behind the selection)
        self._pending = None
        # decode handles (csr_from_selection(..., t_src_dev=...)): the device counter the row widths follow; T_src is then the
        # fixed capacity the column ids are encoded with
        self.t_src_dev = None

    @property
    def col(self):
        if self._pending is not None:
            emit, self._pending = self._pending[3], None
            emit()
        return self._col

    @property
    def col_is_pending(self) -> bool:
        return self._pending is not None

    @property
    def shape(self):
        return (self.N, self.T_dst, self.H * self.T_src)

    @property
    def is_sparse_csr(self):  # duck-typing for callers that only branch on the layout
        return True

    @property
    def device(self):
        return self.crow.device

    def nnz(self) -> torch.Tensor:
        """device tensor (N,) of valid entries per batch item (no host sync)."""
        return self.crow[:, -1]

    def items(self, n0: int, n1: int) -> "FlatCSR":
        """Batch items n0 .. n1-1 as a handle of their own over the same storage (every index tensor is sliced along its
        leading dimension: no copy).  A launch over the part (e.g. one chunk of a chunked all-gather pipeline,
        distributed.ChunkedContextGatherer) writes / reads its rows of the shared arrays.  While the parent's columns are
        pending the part's are too: the fused attention launch writes the part's columns; any other reader of the part's
        `.col` runs the parent's emit launch (the whole batch)."""
        cut = lambda t: t[n0:n1] if t is not None else None
        sub = FlatCSR(self.crow[n0:n1], self._col[n0:n1], self.head_off[n0:n1], self.H, self.T_src, cut(self.bits), cut(self.row_nnz),
                      cut(self.vals))
        sub.t_src_dev = self.t_src_dev
        if self._pending is not None:
            T_m, k, causal, _emit = self._pending
            sub._pending = (T_m, k, causal, lambda: self.col)
        return sub

    def with_values(self, vals: torch.Tensor) -> "FlatCSR":
        """Same structure (shared index tensors), other values."""
        return FlatCSR(self.crow, self.col, self.head_off, self.H, self.T_src, self.bits, self.row_nnz, vals)

    def to_sparse_csr(self, values: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Materialise the reference's wire format: int64 indices, trimmed to Z = max nnz
        (one host sync, like `Z = ....max().item()` at causal_resize_m_to_t.py:667)."""
        Z = int(self.crow[:, -1].max().item())
        crow = self.crow.to(torch.int64)
        col = self.col[:, :Z].to(torch.int64)
        valid = torch.arange(Z, device=col.device).view(1, -1) < crow[:, -1:]
        col = col * valid                                   # the reference zero-fills the padding
        values = values if values is not None else self.vals
        if values is None:
            values = torch.ones((self.N, Z), dtype=torch.float32, device=col.device)
        else:
            values = values[:, :Z] * valid
        return torch.sparse_csr_tensor(crow, col, values, size=self.shape)

    def _wire_format(self):
        if self._wire is None:
            self._wire = self.to_sparse_csr()
        return self._wire

    def crow_indices(self):
        return self._wire_format().crow_indices()

    def col_indices(self):
        return self._wire_format().col_indices()

    def values(self):
        return self._wire_format().values()

    def to(self, device):
        """PerlinAttentionOutput.to(device) moves every field (attention.py:95-106)."""
        mv = lambda t: t.to(device) if t is not None else None
        return FlatCSR(mv(self.crow), mv(self.col), mv(self.head_off), self.H, self.T_src, mv(self.bits), mv(self.row_nnz),
                       mv(self.vals))


def keep_table_causal(H, T_dst, T_m, k, k_oversample=1.0, device=None) -> torch.Tensor:
    """K_t of the module path as int32 (T_dst,), computed with the reference's own fp32 expression
    (attention.py:800,849,856,866): clamp_min(round(H * (k*os*T_m / arange(1..T))), 1)."""
    ctl = torch.arange(1, T_dst + 1, dtype=torch.long)
    per = H * (k * k_oversample * T_m / ctl)
    per = torch.clamp_min(torch.round(per), 1)
    per = torch.clamp_max(per, H * T_m)
    out = per.to(torch.int32)
    return out.to(device) if device is not None else out


def keep_table_kernel_test(H, T_dst, T_m, k, device=None) -> torch.Tensor:
    """K_t of the reference's kernel self-tests (causal_topk_masking.py:31,37)."""
    ctl = torch.arange(1, T_dst + 1, dtype=torch.long)
    per = torch.clamp(H * torch.floor(k * T_m / ctl), 1, H * T_m)
    out = torch.clamp_min(per, 1).to(torch.int32)
    return out.to(device) if device is not None else out


def z_capacity(keep_cpu: torch.Tensor, H, T_dst, T_src, T_m, max_k, is_causal=True) -> int:
    """Analytic upper bound of entries per batch item (no device work): row t emits at most
    min(K_t * min(ceil(w_t/T_m), max_k), H * min(w_t, T_m*max_k)) entries."""
    w = torch.arange(1, T_src + 1)[-T_dst:] if is_causal else torch.full((T_dst,), T_src)
    per_pixel = torch.clamp_max(torch.div(w + T_m - 1, T_m, rounding_mode="floor"), max_k)
    kt = keep_cpu.view(-1, T_dst).max(0).values.to(torch.long)
    bound = torch.minimum(kt * per_pixel, H * torch.minimum(w, torch.tensor(T_m * max_k)))
    return max(int(bound.sum().item()), 1)


# ------------------------------------------------------------------------------------------------
# fused path: probs -> FlatCSR
# ------------------------------------------------------------------------------------------------
@_lib.device_guarded
def topk_to_csr(probs: torch.Tensor, keep: torch.Tensor, k: int, target_width: Optional[int] = None,
                is_causal: bool = True, z_cap: Optional[int] = None, want_mask: bool = False, defer_emit: bool = False):
    """Grouped top-k + nearest-neighbour interpolation to a FlatCSR in three launches, no host sync.

    probs (N,H,T_dst,T_m) f32/f16/bf16 (pixel stride 1); keep int32 device tensor (T_dst,) or (N,T_dst).
    Returns (FlatCSR, mask or None) where mask is the 0/1 fp32 (N,H,T_dst,T_m) compressed mask
    (`partial_attention_mask_before_interp`).  `defer_emit`: leave the column array pending (see `csr_from_selection`) -- the
    fused attention launch expands the kept pixels itself; two launches here instead of three.
    Replaces attention.py:774-947 + ops/kernels/causal_resize_m_to_t.py:910-1007.
    """
    lib = _lib.load()
    probs = getattr(probs, "materialize", lambda: probs)()      # a lazily produced map (ops.LazyTensor) is computed here
    _lib.require_gpu(probs, keep)
    assert probs.ndim == 4 and probs.stride(-1) == 1
    N, H, T_dst, T_m = probs.shape
    T_src = target_width if target_width is not None else T_dst
    assert keep.dtype == torch.int32 and keep.is_contiguous()
    assert keep.shape in ((T_dst,), (N, T_dst))
    keep_stride_n = T_dst if keep.ndim == 2 else 0
    dev = probs.device
    W = (H * T_m + 31) // 32
    bits = torch.empty((N, T_dst, W), dtype=torch.int32, device=dev)
    row_nnz = torch.empty((N, T_dst), dtype=torch.int32, device=dev)
    head_off = torch.empty((N, T_dst, H + 1), dtype=torch.int32, device=dev)
    mask = torch.empty((N, H, T_dst, T_m), dtype=torch.float32, device=dev) if want_mask else None
    st = _lib.stream_ptr()
    _lib.check(lib.sea_topk_select(
        _p(probs), _lib.dtype_code(probs.dtype), N, H, T_dst, T_m,
        probs.stride(0), probs.stride(1), probs.stride(2),
        _p(keep), keep_stride_n, T_src, int(is_causal), int(k),
        _p(bits), _p(mask), _p(row_nnz), _p(head_off), st), "sea_topk_select")
    return csr_from_selection(bits, row_nnz, head_off, H, T_m, T_src, int(k), is_causal, z_cap, keep, defer_emit=defer_emit), mask


@_lib.device_guarded
def csr_from_selection(bits: torch.Tensor, row_nnz: torch.Tensor, head_off: torch.Tensor, H: int, T_m: int, T_src: int,
                       k: int, is_causal: bool = True, z_cap: Optional[int] = None, keep: Optional[torch.Tensor] = None,
                       t_src_dev: Optional[torch.Tensor] = None, defer_emit: bool = False, crow: Optional[torch.Tensor] = None):
    """Row scan + emit: the (bits, row_nnz, head_off) of a selection launch (sea_topk_select or the fused
    sea_predictor_tail_select) -> FlatCSR.  Two launches, no host sync.
    Decode form (`sea_csr_emit_at`): `t_src_dev` (one int32 on the device) is the sequence length the row widths follow,
    `T_src` the FIXED capacity the column ids are encoded with (the FlatCSR says T_src = capacity)."""
    lib = _lib.load()
    N, T_dst = row_nnz.shape
    dev = bits.device
    st = _lib.stream_ptr()
    if crow is None:
        crow = torch.empty((N, T_dst + 1), dtype=torch.int32, device=dev)
        _lib.check(lib.sea_csr_row_scan(_p(row_nnz), N, T_dst, _p(crow), 4, st), "sea_csr_row_scan")
    else:                     # the selection launch already wrote it (one row per item: sea_predictor_tail_select_at's crow_out)
        assert crow.dtype == torch.int32 and tuple(crow.shape) == (N, T_dst + 1) and crow.is_contiguous()
    if z_cap is None:
        z_cap = z_capacity(keep.cpu(), H, T_dst, T_src, T_m, int(k), is_causal)
    col = torch.empty((N, z_cap), dtype=torch.int32, device=dev)
    if t_src_dev is not None:
        assert t_src_dev.dtype == torch.int32 and t_src_dev.numel() == 1 and t_src_dev.is_cuda
        def emit_at():
            with torch.cuda.device(dev):
                _lib.check(lib.sea_csr_emit_at(
                    _p(bits), _p(crow), N, H, T_dst, T_m, _p(t_src_dev), T_src, int(is_causal), int(k),
                    _p(col), 4, col.stride(0), z_cap, _lib.stream_ptr()), "sea_csr_emit_at")
        csr = FlatCSR(crow, col, head_off, H, T_src, bits=bits, row_nnz=row_nnz)
        csr.t_src_dev = t_src_dev
        if defer_emit:      # sparse_attention's decode form (sea_sparse_attention_fused_at) expands the kept pixels itself
            csr._pending = (int(T_m), int(k), bool(is_causal), emit_at)
        else:
            emit_at()
        return csr
    def emit():
        with torch.cuda.device(dev):
            _lib.check(lib.sea_csr_emit(
                _p(bits), _p(crow), _p(head_off), N, H, T_dst, T_m, T_src, int(is_causal), int(k),
                _p(col), 4, col.stride(0), z_cap, None, _lib.stream_ptr()), "sea_csr_emit")
    csr = FlatCSR(crow, col, head_off, H, T_src, bits=bits, row_nnz=row_nnz)
    if defer_emit:          # the fused attention launch (sparse_attention(..., fuse_emit=True)) will write `col`; any other
        csr._pending = (int(T_m), int(k), bool(is_causal), emit)    # reader of `.col` triggers this emit launch itself
    else:
        emit()
    return csr


@_lib.device_guarded
def topk_mask(probs: torch.Tensor, keep: torch.Tensor, k: int, target_width=None, is_causal=True) -> torch.Tensor:
    """a6 alone: 0/1 fp32 compressed mask (N,H,T_dst,T_m)."""
    lib = _lib.load()
    probs = getattr(probs, "materialize", lambda: probs)()
    _lib.require_gpu(probs, keep)
    N, H, T_dst, T_m = probs.shape
    T_src = target_width if target_width is not None else T_dst
    dev = probs.device
    W = (H * T_m + 31) // 32
    bits = torch.empty((N, T_dst, W), dtype=torch.int32, device=dev)
    row_nnz = torch.empty((N, T_dst), dtype=torch.int32, device=dev)
    head_off = torch.empty((N, T_dst, H + 1), dtype=torch.int32, device=dev)
    mask = torch.empty((N, H, T_dst, T_m), dtype=torch.float32, device=dev)
    keep_stride_n = T_dst if keep.ndim == 2 else 0
    _lib.check(lib.sea_topk_select(
        _p(probs), _lib.dtype_code(probs.dtype), N, H, T_dst, T_m,
        probs.stride(0), probs.stride(1), probs.stride(2),
        _p(keep), keep_stride_n, T_src, int(is_causal), int(k),
        _p(bits), _p(mask), _p(row_nnz), _p(head_off), _lib.stream_ptr()), "sea_topk_select")
    return mask


_PATHS = {"auto": _lib.SEA_ATTN_AUTO, "gather": _lib.SEA_ATTN_GATHER, "tile": _lib.SEA_ATTN_TILE}


_FEW_ROWS = None


def attention_few_rows() -> int:
    """(n, h, t) rows up to which the attention launch gives every row a whole wave (a decoding step) -- `sea_attention_few_rows`."""
    global _FEW_ROWS
    if _FEW_ROWS is None:
        _FEW_ROWS = int(_lib.load().sea_attention_few_rows())
    return _FEW_ROWS


def fused_interp_supported(dtype, D: int, T_m: int, rows: int = None) -> bool:
    """Shapes `sea_sparse_attention_fused` covers (steps I + J in one launch): rows of 8 or 16 lanes (16-bit d = 64 / 80 / 128,
    fp32 d = 32 / 64), T_m a multiple of 32 -- and, when the caller says how many (n, h, t) rows the launch has, more than
    `attention_few_rows()` of them (a decoding step runs emit + the wave-per-row kernel instead)."""
    if rows is not None and rows <= attention_few_rows():
        return False
    vec = 4 if dtype == torch.float32 else 8
    lanes = 1
    while lanes * vec < D:
        lanes *= 2
    return lanes in (8, 16) and D % vec == 0 and T_m % 32 == 0


_BWD_WS = {}


def _bwd_workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch of the gather backward (16 B per CSR slot + two int32 per (head, key)): one buffer per (device, stream), grown
    on demand and re-used by every layer's backward on that stream (they run one after another there; the contents are
    undefined between calls) instead of a fresh ~1 GB allocation per layer at OPT-1.3B x 8.  Keyed by the
    stream as well: two backward passes on different streams (or threads) of one device never share scratch, so the caching
    allocator's stream ordering is not needed.  While a HIP graph is being captured the buffer is NOT cached (a block
    allocated then belongs to the graph's private pool: a global must not point into it) -- the capture gets its own."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((nbytes,), dtype=torch.uint8, device=device)
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _BWD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _BWD_WS[key] = ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    return ws[:nbytes]


def clear_bwd_workspace():
    """Release the cached backward scratch (about 1 GB per device and stream at OPT-1.3B x 8); `clear_prep_cache()` calls it."""
    _BWD_WS.clear()


class _SparseAttentionFn(torch.autograd.Function):
    """o = sum_e softmax_e(q . k_e) v_e over the flat CSR, with a backward on the HIP kernels.

    Forward: the gather kernels with the per-entry probabilities kept (they ARE the saved activations: one fp32 per entry
    instead of the dense branch's (N,H,T,T) probability tensor, attention.py:1120-1128).  Backward:
    `sea_sparse_attention_bwd_gather` (csrc/sea_attn_bwd.hip) -- dQ by rows, dK / dV gathered over the transposed pattern;
    `backward_form = "atomic"` selects the first version (`sea_sparse_attention_bwd`: dK / dV by fp32 atomics), which also
    serves rows wider than 16 lanes.  Row scale and the average mix stay outside (plain torch ops on the result), so
    autograd owns their gradients."""
    backward_form = "gather"

    @staticmethod
    def forward(ctx, q, k, v, csr):
        with torch.no_grad():
            out, probs = sparse_attention(q.detach(), k.detach(), v.detach(), csr, want_probs=True, path="gather")
        ctx.save_for_backward(q, k, v, probs, out)
        ctx.csr = csr
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, probs, out = ctx.saved_tensors
        csr = ctx.csr
        lib = _lib.load()
        N, H, T_dst, D = q.shape
        T_src = k.shape[2]
        qd, kd, vd = (t.detach() if t.stride(-1) == 1 else t.detach().contiguous() for t in (q, k, v))
        dout = dout.to(torch.float32).contiguous()
        dq = torch.empty((N, H, T_dst, D), dtype=torch.float32, device=q.device)
        vec = 4 if q.dtype == torch.float32 else 8
        gather = _SparseAttentionFn.backward_form != "atomic" and D <= 16 * vec
        with torch.cuda.device(q.device):
            if gather:
                # dK / dV gathered over the transposed pattern (no float atomics): every row is written by the column pass
                dk = torch.empty((N, H, T_src, D), dtype=torch.float32, device=q.device)
                dv = torch.empty((N, H, T_src, D), dtype=torch.float32, device=q.device)
                nb = int(lib.sea_sparse_attention_bwd_workspace_bytes(N, H, T_src, csr.col.stride(0)))
                ws = _bwd_workspace(nb, q.device)
                _lib.check(lib.sea_sparse_attention_bwd_gather(
                    _p(qd), _p(kd), _p(vd), _lib.dtype_code(q.dtype), N, H, T_dst, T_src, D,
                    _lib.strides3(qd), _lib.strides3(kd), _lib.strides3(vd),
                    _p(csr.crow), _p(csr.col), csr.col.stride(0), _p(csr.head_off),
                    _p(probs), probs.stride(0), _p(out), _p(dout), _p(dq), _p(dk), _p(dv), _p(ws), nb, _lib.stream_ptr()),
                    "sea_sparse_attention_bwd_gather")
            else:
                dk = torch.zeros((N, H, T_src, D), dtype=torch.float32, device=q.device)
                dv = torch.zeros((N, H, T_src, D), dtype=torch.float32, device=q.device)
                _lib.check(lib.sea_sparse_attention_bwd(
                    _p(qd), _p(kd), _p(vd), _lib.dtype_code(q.dtype), N, H, T_dst, T_src, D,
                    _lib.strides3(qd), _lib.strides3(kd), _lib.strides3(vd),
                    _p(csr.crow), _p(csr.col), csr.col.stride(0), _p(csr.head_off),
                    _p(probs), probs.stride(0), _p(out), _p(dout), _p(dq), _p(dk), _p(dv), _lib.stream_ptr()),
                    "sea_sparse_attention_bwd")
        return dq.to(q.dtype), dk.to(k.dtype), dv.to(v.dtype), None


def sparse_attention_autograd(q, k, v, csr: FlatCSR, row_scale=None, avg=None, mix=None) -> torch.Tensor:
    """Differentiable form of `sparse_attention` (fp32 (N,H,T_dst,D) result): the HIP forward + backward for the sparse
    product, torch ops for `* row_scale` and the mix with `avg` (their gradients come from autograd)."""
    o = _SparseAttentionFn.apply(q, k, v, csr)
    if row_scale is not None:
        o = o * row_scale.unsqueeze(-1)
    if mix is not None:
        a = mix.unsqueeze(-1)
        o = o * a + (1.0 - a) * avg.to(o.dtype)
    return o


@_lib.device_guarded
def attention_plan(csr: FlatCSR, T_m: int, is_causal: bool = True, entries_per_tile: float = 0.0):
    """Kernel-choice plan (`sea_attention_plan`): a flat uint8 buffer -- N*H*ceil(T_dst/16) bytes, 1 = the block's entries per
    staged tile favour the MFMA tile kernel, followed (4-byte aligned) by the int32 count of such blocks
    (`plan_blocks(plan, N, H, T_dst)` views the bytes as (N, H, blocks)).  `sparse_attention(path="auto", plan=...)` launches
    both kernels and the count decides on the device which ONE runs the launch.  Needs the kept-pixel bit masks of the
    selection (`csr.bits`); returns None when there are none (a CSR that did not come from topk_to_csr / csr_from_selection)
    or the shape is outside the plan kernel (T_m % 32, H <= 64)."""
    if csr.bits is None or T_m % 32 != 0 or csr.H > 64 or csr.H * T_m > 32768:
        return None
    lib = _lib.load()
    _lib.require_gpu(csr.bits)
    nb = csr.N * csr.H * ((csr.T_dst + 15) // 16)
    buf = torch.empty((((nb + 3) & ~3) + 4,), dtype=torch.uint8, device=csr.bits.device)   # bytes + the int32 tile-block count
    _lib.check(lib.sea_attention_plan(_p(csr.bits), csr.N, csr.H, csr.T_dst, csr.T_src, int(T_m), int(is_causal),
                                      float(entries_per_tile), _p(buf), _lib.stream_ptr()), "sea_attention_plan")
    return buf


@_lib.device_guarded
def sparse_attention(q, k, v, csr: FlatCSR, row_scale: Optional[torch.Tensor] = None,
                     avg: Optional[torch.Tensor] = None, mix: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                     path: str = "auto", want_probs: bool = False, row_tiles: int = 0, key_window: int = 0,
                     plan: Optional[torch.Tensor] = None, fuse_emit: bool = True, keep_columns_pending: bool = False):
    """Fused SDDMM + per-(row,head) softmax + row scale + SpMM (+ mix) over the flat CSR (`sea_sparse_attention_ex`).

    q (N,H,T_dst,D), k/v (N,H,T_src,D), any [n,h,t] strides, feature stride 1.
    row_scale, mix: fp32 (N,H,T_dst) contiguous (already passed through sigmoid); avg like v.
    out: optional preallocated tensor viewed as (N,H,T_dst,D) with arbitrary strides -- pass a
    permuted view of an (N,T_dst,H*D) buffer to get the layout of attention.py:1279-1282 directly.
    Default output: fp32 (N,H,T_dst,D) (flat_csr_sdbmm.py:347 returns fp32).
    path: "auto" | "gather" (row-indexed gather kernels) | "tile" (MFMA tile kernel: 16-bit data, D in {64,80,128});
    row_tiles / key_window tune the tile kernel (0 = defaults).  The tile kernel (also inside "auto" with a plan) stages whole
    16-key tiles: every V row below T_src must be finite, kept or not (0 * Inf = NaN; include/sea_hip.h); the gather kernels
    read kept keys only.
    plan: with path="auto", the kernel choice of `attention_plan` (both kernels are launched, the plan's count of
    tile-favouring blocks decides on the device which one runs; the other exits at once); without it "auto" means the
    gather kernels.
    want_probs: also return the per-entry values rs * softmax (fp32, laid out like csr.col) -- what the reference
    hands out as `partial_attention_probs` (attention.py:1162-1171); returns (out, probs).
    keep_columns_pending: with a handle whose columns are pending and the fused launch serving it, do NOT write the column
    array (the launch keeps the expanded columns in LDS): the handle stays pending and whoever reads `.col` later runs the
    emit launch -- for callers that return the CSR without anybody reading it (the layer's hot path)."""
    lib = _lib.load()
    _lib.require_gpu(q, k, v, csr.crow)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (q, k, v, row_scale, avg, mix)):
        assert out is None and not want_probs, "the differentiable form returns a new fp32 tensor"
        return sparse_attention_autograd(q, k, v, csr, row_scale, avg, mix)
    N, H, T_dst, D = q.shape
    # a handle whose columns are still pending (csr_from_selection(..., defer_emit=True)): the gather kernels expand the
    # kept pixels themselves and WRITE `col` (sea_sparse_attention_fused) where their fused form exists; anything else
    # (the tile kernel, a plan that may choose it, rows of 4 lanes / d = 80 / wider than 16 lanes) reads `.col`, which
    # runs the emit launch first
    decode_form = csr.t_src_dev is not None                   # a decoding step: the sequence length lives in device memory
    fused = (csr.col_is_pending and fuse_emit and path != "tile" and not (path == "auto" and plan is not None)
             and fused_interp_supported(q.dtype, D, csr._pending[0], None if decode_form else N * H * T_dst)
             and not (decode_form and (want_probs or T_dst > 8)))
    T_src = k.shape[2]
    assert k.shape == (N, H, T_src, D) and v.shape == (N, H, T_src, D)
    assert q.dtype == k.dtype == v.dtype
    assert (csr.N, csr.T_dst, csr.H, csr.T_src) == (N, T_dst, H, T_src)
    if out is None:
        out = torch.empty((N, H, T_dst, D), dtype=out_dtype or torch.float32, device=q.device)
    assert out.shape == (N, H, T_dst, D) and out.stride(-1) == 1
    if row_scale is not None:
        assert row_scale.dtype == torch.float32 and row_scale.shape == (N, H, T_dst) and row_scale.is_contiguous()
    if mix is not None:
        assert avg is not None and avg.shape == (N, H, T_dst, D) and avg.dtype == q.dtype
        assert mix.dtype == torch.float32 and mix.shape == (N, H, T_dst) and mix.is_contiguous()
    if plan is not None:
        nb_ = N * H * ((T_dst + 15) // 16)
        assert plan.dtype == torch.uint8 and plan.numel() == ((nb_ + 3) & ~3) + 4 and plan.is_contiguous()
    if fused:
        T_m_, max_k_, causal_, _emit = csr._pending
        raw_col = csr._col
        probs = torch.zeros(raw_col.shape, dtype=torch.float32, device=q.device) if want_probs else None
        common = (_p(q), _p(k), _p(v), _lib.dtype_code(q.dtype), N, H, T_dst, T_src, D,
                  _lib.strides3(q), _lib.strides3(k), _lib.strides3(v),
                  _p(csr.crow), _p(raw_col), raw_col.stride(0), _p(csr.head_off),
                  _p(row_scale), _p(avg), _lib.strides3(avg) if avg is not None else None, _p(mix),
                  _p(out), _lib.dtype_code(out.dtype), _lib.strides3(out))
        write_cols = 0 if keep_columns_pending else 1
        if decode_form:
            rc = lib.sea_sparse_attention_fused_at(*common, _p(csr.bits), T_m_, _p(csr.t_src_dev), int(causal_), max_k_, write_cols,
                                                   _lib.stream_ptr())
        else:
            rc = lib.sea_sparse_attention_fused(*common, _p(probs), probs.stride(0) if probs is not None else 0,
                                                _p(csr.bits), T_m_, int(causal_), max_k_, write_cols, _lib.stream_ptr())
        if rc == 0:
            if not keep_columns_pending:
                csr._pending = None                         # the launch has written the columns
            return (out, probs) if want_probs else out
        if rc != _lib.SEA_EUNSUPPORTED:
            _lib.check(rc, "sea_sparse_attention_fused")
        # a shape the fused form does not cover (nothing was launched): emit, then the plain operator below
    probs = torch.zeros(csr.col.shape, dtype=torch.float32, device=q.device) if want_probs else None
    flags = _PATHS[path] | ((int(row_tiles) & 0xf) << 8)
    if key_window:
        assert key_window & (key_window - 1) == 0, "key_window is a power of two"
        flags |= (int(key_window).bit_length() - 1) << 12
    _lib.check(lib.sea_sparse_attention_ex(
        _p(q), _p(k), _p(v), _lib.dtype_code(q.dtype), N, H, T_dst, T_src, D,
        _lib.strides3(q), _lib.strides3(k), _lib.strides3(v),
        _p(csr.crow), _p(csr.col), csr.col.stride(0), _p(csr.head_off),
        _p(row_scale), _p(avg), _lib.strides3(avg) if avg is not None else None, _p(mix),
        _p(out), _lib.dtype_code(out.dtype), _lib.strides3(out),
        _p(probs), probs.stride(0) if probs is not None else 0,
        _p(plan) if (plan is not None and path == "auto" and not want_probs) else c_void_p(0), flags,
        _lib.stream_ptr()), "sea_sparse_attention")
    return (out, probs) if want_probs else out


def plan_blocks(plan: torch.Tensor, N: int, H: int, T_dst: int) -> torch.Tensor:
    """(N, H, ceil(T_dst/16)) view of a dispatch plan's per-block bytes."""
    tb = (T_dst + 15) // 16
    return plan[:N * H * tb].view(N, H, tb)


def make_plan(blocks: torch.Tensor) -> torch.Tensor:
    """A dispatch plan buffer from explicit per-block choices (uint8 (N, H, blocks)); tests and experiments."""
    nb = blocks.numel()
    buf = torch.zeros((((nb + 3) & ~3) + 4,), dtype=torch.uint8, device=blocks.device)
    buf[:nb] = blocks.reshape(-1)
    buf[-4:].view(torch.int32)[0] = int(blocks.sum().item()) if blocks.device.type == "cpu" else blocks.sum().to(torch.int32)
    return buf


def sparse_attention_bytes(Z: int, N: int, H: int, T_dst: int, D: int, elem_bytes: int) -> int:
    """Algorithmic bytes of one fused launch."""
    return int(_lib.load().sea_sparse_attention_bytes(Z, N, H, T_dst, D, elem_bytes))


# ------------------------------------------------------------------------------------------------
# drop-in operators (ops/__init__.py:1-7)
# ------------------------------------------------------------------------------------------------
def _csr_parts(t: torch.Tensor):
    assert t.is_sparse_csr
    crow, col, val = t.crow_indices(), t.col_indices(), t.values()
    assert crow.dtype == torch.int64 and col.dtype == torch.int64
    return crow.contiguous(), col.contiguous(), val


@_lib.device_guarded
def resize_from_m_to_t_csr(x, masked_fill_value, k, target_width=None, training=False, need_assert=False,
                           is_causal=True, max_col_z=None, benchmarking=False, oversampled=None):
    """Drop-in for ops/kernels/causal_resize_m_to_t.py:910-1007 (METHOD 1, scan_col :631-762).

    x: (N,H,T_dst,T_m) 0/1 mask.  Returns torch.sparse_csr_tensor (N, T_dst, H*T_src), int64 indices,
    values = ones of x.dtype, Z = max over the batch (shorter items zero padded) -- one host sync
    for Z exactly like the reference (`.max().item()`, :667)."""
    assert not training
    assert masked_fill_value == 0
    lib = _lib.load()
    _lib.require_gpu(x)
    N, H, T_dst, T_m = x.shape
    T_src = target_width if target_width is not None else T_dst
    if x.stride(-1) != 1 or any(s % 4 for s in x.stride()[:3]):
        x = x.contiguous()
    dev = x.device
    W = (H * T_m + 31) // 32
    bits = torch.empty((N, T_dst, W), dtype=torch.int32, device=dev)
    row_nnz = torch.empty((N, T_dst), dtype=torch.int32, device=dev)
    head_off = torch.empty((N, T_dst, H + 1), dtype=torch.int32, device=dev)
    st = _lib.stream_ptr()
    _lib.check(lib.sea_mask_to_bits(
        _p(x), _lib.dtype_code(x.dtype), N, H, T_dst, T_m, x.stride(0), x.stride(1), x.stride(2),
        T_src, int(is_causal), int(k), _p(bits), _p(row_nnz), _p(head_off), st), "sea_mask_to_bits")
    crow = torch.empty((N, T_dst + 1), dtype=torch.int64, device=dev)
    _lib.check(lib.sea_csr_row_scan(_p(row_nnz), N, T_dst, _p(crow), 8, st), "sea_csr_row_scan")
    Z = int(crow[:, -1].max().item())
    col = torch.zeros((N, Z), dtype=torch.int64, device=dev)
    values = torch.ones((N, Z), dtype=x.dtype, device=dev)
    if Z > 0:
        _lib.check(lib.sea_csr_emit(
            _p(bits), _p(crow), _p(head_off), N, H, T_dst, T_m, T_src, int(is_causal), int(k),
            _p(col), 8, col.stride(0), Z, None, st), "sea_csr_emit")
    return torch.sparse_csr_tensor(crow, col, values, size=(N, T_dst, H * T_src))


@_lib.device_guarded
def flat_csr_masked_bmm(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor, max_z_per_row: int = None):
    """Drop-in for flat_csr_masked_bmm.py:137-195 (SDDMM).  max_z_per_row is accepted and unused:
    the HIP kernel walks each row's own length, so the reference's `.item()` sync (:162-164) is gone."""
    assert mask.is_sparse_csr
    assert a.ndim == b.ndim
    assert a.ndim == 4
    N, H, T_DST, HID = a.shape
    assert b.shape[:2] == (N, H)
    _, _, T_SRC, HID = b.shape
    assert mask.shape == (N, T_DST, H * T_SRC)
    lib = _lib.load()
    _lib.require_gpu(a, b, mask)
    crow, col, val = _csr_parts(mask)
    out_values = val.to(torch.float32).clone()
    assert crow.shape[0] == N
    if a.stride(-1) != 1:
        a = a.contiguous()
    if b.stride(-1) != 1:
        b = b.contiguous()
    if a.dtype != b.dtype:
        b = b.to(a.dtype)
    if col.shape[1] > 0:
        _lib.check(lib.sea_csr_sddmm(
            _p(a), _p(b), _lib.dtype_code(a.dtype), N, H, T_DST, T_SRC, HID,
            _lib.strides3(a), _lib.strides3(b), _p(crow), _p(col), 8, col.stride(0),
            _p(out_values), _lib.stream_ptr()), "sea_csr_sddmm")
    return torch.sparse_csr_tensor(crow, col, out_values, size=mask.shape)


@_lib.device_guarded
def flat_csr_softmax(scores: torch.Tensor, H: int, T_SRC: int, max_z_per_row: int = None):
    """Drop-in for flat_csr_softmax.py:127-176."""
    assert scores.is_sparse_csr
    lib = _lib.load()
    _lib.require_gpu(scores)
    crow, col, val = _csr_parts(scores)
    in_values = val.to(torch.float32).contiguous()
    out_values = in_values.clone()
    N, R_1 = crow.shape
    if col.shape[1] > 0:
        _lib.check(lib.sea_csr_softmax(
            _p(in_values), _p(out_values), N, H, R_1 - 1, T_SRC, _p(crow), _p(col), 8, col.stride(0),
            _lib.stream_ptr()), "sea_csr_softmax")
    return torch.sparse_csr_tensor(crow, col, out_values, size=scores.shape)


@_lib.device_guarded
def flat_csr_elmul(probs: torch.Tensor, dense: torch.Tensor, max_z_per_row: int = None):
    """Drop-in for flat_csr_elmul.py:110-162.  `dense` may be a stride-0 expanded view
    (the module passes one, attention.py:1170-1171)."""
    assert probs.is_sparse_csr
    N, T_DST, H_T = probs.shape
    _N, H, _T_DST, T = dense.shape
    assert T_DST == _T_DST
    assert N == _N
    assert H_T == H * T
    lib = _lib.load()
    _lib.require_gpu(probs, dense)
    crow, col, val = _csr_parts(probs)
    in_values = val.to(torch.float32).contiguous()
    out_values = in_values.clone()
    if col.shape[1] > 0:
        _lib.check(lib.sea_csr_elmul(
            _p(in_values), _p(out_values), _p(dense), _lib.dtype_code(dense.dtype), _lib.strides4(dense),
            N, H, T_DST, T, _p(crow), _p(col), 8, col.stride(0), _lib.stream_ptr()), "sea_csr_elmul")
    return torch.sparse_csr_tensor(crow, col, out_values, size=probs.shape)


@_lib.device_guarded
def flat_csr_sdbmm(scores: torch.Tensor, value_layer: torch.Tensor, T_M: int, max_z_per_row: int = None,
                   benchmarking: bool = False):
    """Drop-in for flat_csr_sdbmm.py:323-439 (SpMM).  Output fp32 (N,H,T_dst,D) like the reference
    (:347).  Rows must be grouped by ascending head (the reference relies on the same, :227-263);
    unlike the reference no entry is dropped when a head holds more than MAX_ROW_T entries (:382-388)."""
    assert scores.is_sparse_csr
    lib = _lib.load()
    _lib.require_gpu(scores, value_layer)
    crow, col, val = _csr_parts(scores)
    values = val.to(torch.float32).contiguous()
    other = value_layer
    N, R_1 = crow.shape
    _N, H, T_SRC, HID = other.shape
    assert N == _N
    _N, T_DST, HT_SRC = scores.shape
    assert N == _N
    assert HT_SRC == (H * T_SRC)
    if other.stride(-1) != 1:
        other = other.contiguous()
    output = torch.zeros((N, H, T_DST, HID), device=values.device)
    if col.shape[1] == 0:
        return output
    st = _lib.stream_ptr()
    head_off = torch.empty((N, T_DST, H + 1), dtype=torch.int32, device=values.device)
    _lib.check(lib.sea_csr_head_offsets(
        _p(crow), _p(col), 8, N, H, T_DST, T_SRC, col.stride(0), _p(head_off), st), "sea_csr_head_offsets")
    _lib.check(lib.sea_csr_spmm(
        _p(values), _p(other), _lib.dtype_code(other.dtype), N, H, T_DST, T_SRC, HID, _lib.strides3(other),
        _p(crow), _p(col), 8, col.stride(0), _p(head_off), _p(output), st), "sea_csr_spmm")
    return output


def flat_csr_to_dense(csr, T_SRC, H):
    """Drop-in for flat_csr_to_dense.py:3-36 (debug/parity probe): (N,H,T_dst,T_src) of the valid
    entries.  Accepts a torch CSR tensor or a FlatCSR (then values are ones)."""
    if isinstance(csr, FlatCSR):
        csr = csr.to_sparse_csr()
    assert csr.is_sparse_csr
    N, T_DST, H_T = csr.shape
    crow, col, values = csr.crow_indices(), csr.col_indices(), csr.values()
    Z = col.shape[-1]
    out = torch.zeros((N, H, T_DST, T_SRC), dtype=values.dtype, device=values.device)
    if Z == 0:
        return out
    pos = torch.arange(Z, device=col.device).view(1, Z)
    valid = pos < crow[:, -1:]
    rows = (torch.searchsorted(crow.contiguous(), pos.expand(N, Z).contiguous(), right=True) - 1).clamp_(0, T_DST - 1)
    n_idx = torch.arange(N, device=col.device).view(N, 1).expand(N, Z)
    h_idx = torch.div(col, T_SRC, rounding_mode="floor")
    k_idx = col - h_idx * T_SRC
    out.index_put_((n_idx[valid], h_idx[valid], rows[valid], k_idx[valid]), values[valid], accumulate=True)
    return out
