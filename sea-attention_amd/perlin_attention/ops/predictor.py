"""HIP kernels for the bandwidth-bound pieces of SEA's estimator and epilogue (csrc/sea_predictor.hip).
Not part of the reference's operator package: there these steps are chains of framework kernels inside
`PerlinAttention.forward` (attention.py:123-131,266-281,670-673,1220-1222)."""
from ctypes import c_void_p, c_int64 as ctypes_i64
from typing import Optional

import weakref

import torch

from ... import _lib


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


class LazyTensor(torch.Tensor):
    """A tensor whose values are computed the first time anything needs them.  Shape, dtype, device and strides are
    served from the wrapper; any torch operation on it (indexing, `.float()`, `.cpu()`, comparisons, ...) runs the
    thunk once and proceeds on the real tensor.  Used for `estimated_attention_probs(_m)` in sparse mode: the module
    returns the (N,H,T,T_m) map (attention.py:1343) but nothing on the hot path reads it, and writing it is 537 MB per
    step at OPT-1.3B x 8 -- the fused tail + selection launch keeps it on chip and this handle recomputes it from the
    kept conv output on demand (same kernel code: bit-identical values).
    The thunk reads the layer's weights and the kept intermediate AT FIRST ACCESS, on the stream that is current then: a
    caller that edits the weights in place (or loads a state dict) between the step and the first read gets the map of the
    NEW weights, not the one the selection used -- `ops.realize(t)` first, or run the module with
    `lazy_attention_probs = False` (the reference's eager tensor)."""

    @staticmethod
    def __new__(cls, shape, dtype, device, thunk):
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, device=device, requires_grad=False)
        r._thunk, r._real = thunk, None
        return r

    def __init__(self, shape, dtype, device, thunk):
        pass

    def materialize(self) -> torch.Tensor:
        if self._real is None:
            self._real = self._thunk()
            self._thunk = None
            assert tuple(self._real.shape) == tuple(self.shape) and self._real.dtype == self.dtype
        return self._real

    @property
    def is_materialized(self) -> bool:
        return self._real is not None

    # C-level consumers (ctypes bindings, numpy, pickling) see the real tensor: a wrapper subclass has no storage, its own
    # data_ptr() would be 0 -- and an optional pointer argument would then be skipped without any error
    def data_ptr(self):
        return self.materialize().data_ptr()

    def numpy(self, *a, **kw):
        return self.materialize().numpy(*a, **kw)

    def tolist(self):
        return self.materialize().tolist()

    def __reduce_ex__(self, proto):
        return self.materialize().__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        return self.materialize().clone()

    def __repr__(self):
        return f"LazyTensor(shape={tuple(self.shape)}, dtype={self.dtype}, device={self.device}, materialized={self._real is not None})"

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        un = lambda x: x.materialize() if isinstance(x, LazyTensor) else x
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))


def realize(t):
    """The real tensor behind a LazyTensor (any other value is returned as it is): what goes to the C ABI."""
    return t.materialize() if isinstance(t, LazyTensor) else t


_prep_cache = {}
_prep_generation = 0          # bumped by clear_prep_cache(): holders of raw pointers into cached packs compare it
_prep_recorders = []          # lists that collect every value `_cached` hands out while a `pinned_prep()` block is open


def clear_prep_cache():
    """Drop every cached re-layout of constant weights.  The cache notices new tensors, `.to(device)` moves,
    storage swaps and in-place autograd-visible edits by itself; an edit made behind autograd's back
    (`param.data.mul_()`, `param.data.copy_()`: EMA swaps, pruning) bumps no version counter, so whoever does
    that calls this afterwards.  `PerlinAttention.load_state_dict / _apply` call it (attention.py).  A captured HIP
    graph holds raw pointers into the packs: `DecodeSession` pins the ones its launches used and re-captures when
    `prep_generation()` has moved."""
    global _prep_generation
    _prep_generation += 1
    _prep_cache.clear()
    from .flat_csr import clear_bwd_workspace
    clear_bwd_workspace()


def prep_generation() -> int:
    return _prep_generation


class pinned_prep:
    """`with pinned_prep() as pins:` -- `pins` collects every prepared tensor (tuple) the operators inside the block take from
    the cache or build: keeping the list alive keeps their memory alive, whatever happens to the cache afterwards."""

    def __enter__(self):
        self.values = []
        _prep_recorders.append(self.values)
        return self.values

    def __exit__(self, *exc):
        _prep_recorders.remove(self.values)
        return False


def _cached(tag, tensors, dtype, build):
    """Small re-layouts of constant weights (inference), cached per source tensor.  An entry is valid only while
    the very same tensor objects (weak references to the view bases) are alive, unmodified (`_version`) and still
    sit at the same address on the same device: a freed tensor whose address is handed to a new one of the same
    shape must not hit, and neither may a parameter after `module.to(device)` / `param.data = other`."""
    bases = [t._base if t._base is not None else t for t in tensors]
    key = (tag, dtype) + tuple((id(b), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.data_ptr(), str(t.device))
                               for b, t in zip(bases, tensors))
    hit = _prep_cache.get(key)
    value = None
    if hit is not None:
        refs, versions, val = hit
        if all(r() is b for r, b in zip(refs, bases)) and versions == [b._version for b in bases]:
            value = val
    if value is None:
        if len(_prep_cache) > 256:
            _prep_cache.clear()
        value = build()
        _prep_cache[key] = ([weakref.ref(b) for b in bases], [b._version for b in bases], value)
    for rec in _prep_recorders:
        rec.append(value)
    return value


@_lib.device_guarded
def split_layernorm(x: torch.Tensor, splits: int, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
                    gelu: bool = False):
    """ChannelSplit(splits) followed by LayerNorm over the new (narrower) last dim, optionally followed by
    the exact (erf) GELU.  x (N,C,T,splits*W) -> (N,C*splits,T,W).  splits=1 is a plain row LayerNorm."""
    lib = _lib.load()
    _lib.require_gpu(x, weight, bias)
    N, C, T, SW = x.shape
    assert SW % splits == 0
    W = SW // splits
    x = x.contiguous()
    w, b = _cached("ln", (weight, bias), x.dtype, lambda: (weight.to(x.dtype).contiguous(), bias.to(x.dtype).contiguous()))
    out = torch.empty((N, C * splits, T, W), dtype=x.dtype, device=x.device)
    _lib.check(lib.sea_split_layernorm(_p(x), _lib.dtype_code(x.dtype), N, C, T, splits, W, _p(w), _p(b), float(eps),
                                       int(gelu), _p(out), _lib.stream_ptr()), "sea_split_layernorm")
    return out


@_lib.device_guarded
def predictor_tail(y: torch.Tensor, conv_w: torch.Tensor, conv_b: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor,
                   up: int, T_m: int, eps: float = 1e-5, want_scores: bool = False):
    """upsample(1,up) -> 1x1 conv (pad 1 on width) -> area resize to T_m -> LayerNorm(T_m) -> softmax.
    y (N,C,T,W4) -- or its 5-D C8 form (N,T,C/8,W4,8) -- -> probs (N,H,T,T_m) [, scores].
    conv_w (H,C) is the live row of the causal 1x1 kernel."""
    lib = _lib.load()
    _lib.require_gpu(y, conv_w, conv_b, ln_w, ln_b)
    if y.dim() == 5:
        N, T, C8, W4, _e = y.shape
        assert _e == 8 and y.is_contiguous(), "C8 activations are dense (N, T, C/8, W, 8)"
        C = C8 * 8
    else:
        N, C, T, W4 = y.shape
        if y.stride(-1) != 1 and y.stride(1) != 1:
            y = y.contiguous()
    H = conv_w.shape[0]
    assert conv_w.shape == (H, C) and W4 * up == T_m
    dt = y.dtype
    cw, cb, g, b, w16, Cp = _tail_pack(conv_w, conv_b, ln_w, ln_b, dt, y.device)
    probs = torch.empty((N, H, T, T_m), dtype=dt, device=y.device)
    scores = torch.empty_like(probs) if want_scores else None
    _lib.check(lib.sea_predictor_tail(_p(y), _lib.dtype_code(dt), N, C, H, T, W4, up, T_m, _lib.strides5_blocked(y),
                                      _p(cw), _p(cb), _p(w16), Cp, _p(g), _p(b), float(eps), _p(probs), _p(scores),
                                      _lib.stream_ptr()), "sea_predictor_tail")
    return probs, scores


def _tail_pack(conv_w, conv_b, ln_w, ln_b, dt, device):
    """The cached re-layout of the tail's constants (one entry shared by predictor_tail / _select / _z and conv_z)."""
    H, C = conv_w.shape
    Hpad = (H + 7) // 8 * 8

    def build():
        cw = torch.zeros((C, Hpad), dtype=torch.float32, device=device)  # transposed, head axis padded: scalar-cache reads
        cw[:, :H] = conv_w.to(dt).float().t()
        cb = torch.zeros((Hpad,), dtype=torch.float32, device=device)
        cb[:H] = conv_b.to(dt).float()
        w16, Cp = None, 0
        if dt != torch.float32:                                           # row-major 16-bit copy for the MFMA variant
            Cp, HP = (C + 31) // 32 * 32, (H + 15) // 16 * 16
            w16 = torch.zeros((HP, Cp), dtype=dt, device=device)
            w16[:H, :C] = conv_w.to(dt)
        return cw, cb, ln_w.to(dt).contiguous(), ln_b.to(dt).contiguous(), w16, Cp
    return _cached("tail", (conv_w, conv_b, ln_w, ln_b), dt, build)


def _tail_consts(ln_w: torch.Tensor, ln_b: torch.Tensor, up: int, T_m: int, dt, device):
    """The tail's per-pixel constants (taps of the area resize, gamma, beta: `sea_predictor_tail_consts`) as a device table,
    cached with the LayerNorm weights; None for shapes the table does not serve (T_m != 256, fp32 maps)."""
    if T_m != 256 or dt not in (torch.float16, torch.bfloat16, torch.float32) or T_m % up:
        return None

    def build():
        g, b = ln_w.to(dt).contiguous(), ln_b.to(dt).contiguous()
        tab = torch.empty((3 * 256,), dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            _lib.check(_lib.load().sea_predictor_tail_consts(_lib.dtype_code(dt), T_m // up, up, T_m, _p(g), _p(b), _p(tab),
                                                             _lib.stream_ptr()), "sea_predictor_tail_consts")
        return (tab, g, b)
    return _cached(f"tailtab{up}x{T_m}", (ln_w, ln_b), dt, build)[0]


@_lib.device_guarded
def predictor_tail_z(z: torch.Tensor, conv_w: torch.Tensor, conv_b: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor,
                     up: int, T_m: int, dtype: torch.dtype, eps: float = 1e-5, want_scores: bool = False):
    """predictor_tail from z (N, T, H, W4) fp32 = the 1x1 convolution's output as `causal_conv_c8_z` writes it
    (`sea_predictor_tail_z`): area resize -> LayerNorm -> softmax; probs (N,H,T,T_m) of `dtype` [, scores]."""
    lib = _lib.load()
    _lib.require_gpu(z, conv_w, conv_b, ln_w, ln_b)
    N, T, H, W4 = z.shape
    assert z.dtype == torch.float32 and z.is_contiguous() and conv_w.shape[0] == H and W4 * up == T_m
    assert dtype in (torch.float16, torch.bfloat16)
    _cw, cb, g, b, _w16, _Cp = _tail_pack(conv_w, conv_b, ln_w, ln_b, dtype, z.device)
    probs = torch.empty((N, H, T, T_m), dtype=dtype, device=z.device)
    scores = torch.empty_like(probs) if want_scores else None
    _lib.check(lib.sea_predictor_tail_z(_p(z), _lib.dtype_code(dtype), N, H, T, W4, up, T_m, _p(cb), _p(g), _p(b), float(eps),
                                        _p(probs), _p(scores), _lib.stream_ptr()), "sea_predictor_tail_z")
    return probs, scores


def predictor_tail_select_supported(y: torch.Tensor, H: int, T_m: int, decode: bool = False) -> bool:
    """Shapes csrc/sea_topk.hip: predictor_tail_select(_gen)_kernel take (see sea_predictor_tail_select in sea_hip.h):
    any predictor length T_m % 4 == 0 up to 512 whose row fits the kernel's LDS plan; the decode form (`_at`) and the
    register-resident kernel take T_m = 256 with H % 4 == 0."""
    if y.dtype == torch.float32:      # fp32 data (round 5): the T_m = 256 form on the fp32 MFMA, H <= 32, no decode form
        return (y.dim() == 5 or y.stride(1) == 1) and T_m == 256 and H % 4 == 0 and H <= 32 and not decode
    if not (y.dtype in (torch.float16, torch.bfloat16) and (y.dim() == 5 or y.stride(1) == 1)):
        return False
    if T_m == 256 and H % 4 == 0 and H <= 64:
        return True
    if decode or T_m % 4 or T_m > 512 or H > 64 or H * T_m > 16384:
        return False
    W4, E = T_m // 4, (T_m + 63) // 64
    E = E if E <= 4 else 6 if E <= 6 else 8
    zt = max((H + 15) // 16 * 16 * (W4 + 3) * 4 + 3 * 64 * E * 4, 8192)
    return zt + H * T_m * 2 + 12 * 1024 <= 160 * 1024


@_lib.device_guarded
def predictor_tail_select(y: Optional[torch.Tensor], conv_w: torch.Tensor, conv_b: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor,
                          up: int, T_m: int, keep: torch.Tensor, k: int, T_src: int, is_causal: bool = True,
                          eps: float = 1e-5, want_scores: bool = False, t_src_dev: Optional[torch.Tensor] = None,
                          lazy_probs: bool = False, crow_out: Optional[torch.Tensor] = None,
                          z: Optional[torch.Tensor] = None, map_dtype: Optional[torch.dtype] = None):
    """predictor_tail + grouped top-k selection in one launch.  Returns (probs, scores, (bits, row_nnz, head_off));
    feed the triple to flat_csr.csr_from_selection.  Bit-identical to predictor_tail followed by topk_to_csr.
    `lazy_probs`: the launch does NOT write the (N,H,T,T_m) map (nobody on the hot path reads it); `probs` is then a
    `LazyTensor` that runs `predictor_tail` on the kept `y` (or `predictor_tail_z` on the kept `z`) the first time anything
    touches its values -- with the weights as they are AT THAT MOMENT: realize it (`ops.realize`) before editing them.
    `z` (with `y = None`, `map_dtype`): the 1x1 convolution's output (N, T, H, W4) fp32 from `causal_conv_c8_z`
    (`sea_predictor_tail_select_z`: the row's z tile is a copy instead of loads + MFMAs).
    Decode form (`sea_predictor_tail_select_at`, a step replayed as a HIP graph): `t_src_dev` is a one-element int32
    device tensor holding the sequence length (T_src is ignored) and `keep` a table over absolute row indices."""
    lib = _lib.load()
    H = conv_w.shape[0]
    if z is not None:
        assert y is None and t_src_dev is None and map_dtype in (torch.float16, torch.bfloat16)
        _lib.require_gpu(z, conv_w, conv_b, ln_w, ln_b, keep)
        N, T, Hz, W4 = z.shape
        assert Hz == H and z.dtype == torch.float32 and z.is_contiguous() and W4 * up == T_m
        assert predictor_tail_select_supported(torch.empty((0, 0, 1, 1, 8), dtype=map_dtype), H, T_m)
        dt, dev = map_dtype, z.device
    else:
        _lib.require_gpu(y, conv_w, conv_b, ln_w, ln_b, keep)
        if y.dim() == 5:
            N, T, C8, W4, _e = y.shape
            # rows of C8 blocks; the batch and row strides are free (a decode step hands over the last row of a window: no copy)
            assert _e == 8 and y.stride()[2:] == (W4 * 8, 8, 1) and y.stride(0) % 8 == 0 and y.stride(1) % 8 == 0
            C = C8 * 8
        else:
            N, C, T, W4 = y.shape
        assert conv_w.shape == (H, C) and W4 * up == T_m and predictor_tail_select_supported(y, H, T_m, decode=t_src_dev is not None)
        dt, dev = y.dtype, y.device
    assert keep.dtype == torch.int32 and keep.is_contiguous() and (t_src_dev is not None or keep.shape in ((T,), (N, T)))
    _cw, cb, g, b, w16, Cp = _tail_pack(conv_w, conv_b, ln_w, ln_b, dt, dev)
    if dt == torch.float32:           # fp32 data: the weights go in as the (C, Hpad) fp32 transposed copy, and the map is always written
        w16, Cp, lazy_probs = _cw, 0, False
    tab = _tail_consts(ln_w, ln_b, up, T_m, dt, dev)
    if lazy_probs:
        assert t_src_dev is None
        if z is not None:
            probs = LazyTensor((N, H, T, T_m), dt, dev, lambda: predictor_tail_z(z, conv_w, conv_b, ln_w, ln_b, up, T_m, dt, eps)[0])
        else:
            probs = LazyTensor((N, H, T, T_m), dt, dev, lambda: predictor_tail(y, conv_w, conv_b, ln_w, ln_b, up, T_m, eps)[0])
    else:
        probs = torch.empty((N, H, T, T_m), dtype=dt, device=dev)
    scores = torch.empty((N, H, T, T_m), dtype=dt, device=dev) if want_scores else None
    W = (H * T_m + 31) // 32
    bits = torch.empty((N, T, W), dtype=torch.int32, device=dev)
    row_nnz = torch.empty((N, T), dtype=torch.int32, device=dev)
    head_off = torch.empty((N, T, H + 1), dtype=torch.int32, device=dev)
    assert crow_out is None or (t_src_dev is not None and T == 1 and crow_out.dtype == torch.int32 and tuple(crow_out.shape) == (N, 2)
                                and crow_out.is_contiguous()), "crow_out: the decode form with one row per batch item"
    if t_src_dev is not None:
        assert t_src_dev.dtype == torch.int32 and t_src_dev.numel() == 1 and t_src_dev.is_cuda and keep.ndim == 1
        _lib.check(lib.sea_predictor_tail_select_at(
            _p(y), _lib.dtype_code(dt), N, C, H, T, W4, up, T_m, _lib.strides5_blocked(y), _p(cb), _p(w16), Cp, _p(g), _p(b),
            float(eps), _p(probs), _p(scores), _p(keep), _p(t_src_dev), int(is_causal), int(k),
            _p(bits), _p(row_nnz), _p(head_off), _p(crow_out), _p(tab), _lib.stream_ptr()), "sea_predictor_tail_select_at")
        return probs, scores, (bits, row_nnz, head_off)
    if z is not None:
        _lib.check(lib.sea_predictor_tail_select_z(
            _p(z), _lib.dtype_code(dt), N, H, T, W4, up, T_m, _p(cb), _p(g), _p(b), float(eps),
            _p(None if lazy_probs else probs), _p(scores), _p(keep), T if keep.ndim == 2 else 0, T_src, int(is_causal), int(k),
            _p(bits), _p(row_nnz), _p(head_off), _p(tab), _lib.stream_ptr()), "sea_predictor_tail_select_z")
        return probs, scores, (bits, row_nnz, head_off)
    _lib.check(lib.sea_predictor_tail_select(
        _p(y), _lib.dtype_code(dt), N, C, H, T, W4, up, T_m, _lib.strides5_blocked(y), _p(cb), _p(w16), Cp, _p(g), _p(b),
        float(eps), _p(None if lazy_probs else probs), _p(scores), _p(keep), T if keep.ndim == 2 else 0, T_src, int(is_causal), int(k),
        _p(bits), _p(row_nnz), _p(head_off), _p(tab), _lib.stream_ptr()), "sea_predictor_tail_select")
    return probs, scores, (bits, row_nnz, head_off)


@_lib.device_guarded
def cumavg(v: torch.Tensor, out: Optional[torch.Tensor] = None, n_slices: int = None) -> torch.Tensor:
    """Causal cumulative average over the time axis of (N,H,T,D), fp32 accumulation, dtype preserved.
    `out`: optional preallocated contiguous (N,H,T,D) result (lets a caller launch this on a side stream)."""
    lib = _lib.load()
    _lib.require_gpu(v)
    N, H, T, D = v.shape
    if v.stride(-1) != 1:
        v = v.contiguous()
    if out is None:
        out = torch.empty((N, H, T, D), dtype=v.dtype, device=v.device)
    assert out.shape == (N, H, T, D) and out.dtype == v.dtype and out.is_contiguous()
    # one workgroup per (n, h): with few pairs (a one-sequence-per-GPU shard) the rows are cut into slices
    ns = n_slices if n_slices is not None else (min(16, 256 // (N * H)) if N * H <= 128 and T >= 2048 else 1)
    vec_ok = (v.dtype != torch.float32 and D in (32, 64, 80, 128) and all(st % 8 == 0 for st in v.stride()[:3])
              and v.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0)
    if ns > 1 and vec_ok:
        ws = torch.empty((N * H * ns * D,), dtype=torch.float32, device=v.device)
        _lib.check(lib.sea_cumavg_sliced(_p(v), _lib.dtype_code(v.dtype), N, H, T, D, _lib.strides3(v), _p(out), ns, _p(ws),
                                         ws.numel() * 4, _lib.stream_ptr()), "sea_cumavg_sliced")
        return out
    _lib.check(lib.sea_cumavg(_p(v), _lib.dtype_code(v.dtype), N, H, T, D, _lib.strides3(v), _p(out),
                              _lib.stream_ptr()), "sea_cumavg")
    return out


def performer_supported(D: int, nb: int) -> bool:
    """Shapes the fused Performer kernel is instantiated for (csrc/sea_performer.hip: dispatch_perf)."""
    nbt = (nb + 15) // 16
    return D in (64, 80, 128) and nbt <= 5


def performer_avg_supported(q: torch.Tensor, nb: int) -> bool:
    """Can the Performer launch also emit the cumulative average of v?  Asked of the library (one predicate on both
    sides of the ABI: `sea_performer_avg_supported`)."""
    if q.dtype not in (torch.bfloat16, torch.float16):
        return False
    return bool(_lib.load().sea_performer_avg_supported(int(q.shape[-1]), int(nb), _lib.dtype_code(q.dtype)))


_PLAN_CACHE = {}


def performer_plan(N: int, H: int, T: int, D: int, nb: int, dtype) -> tuple:
    """(n_segments, workspace_bytes) the library proposes for a shape (`sea_performer_plan`): 1 segment when N*H
    workgroups already fill the chip, else the rows are cut so that about 256 workgroups run."""
    key = (N, H, T, D, nb, dtype)
    if key not in _PLAN_CACHE:
        import ctypes
        nseg, ws = (ctypes.c_int64 * 1)(), (ctypes.c_int64 * 1)()
        _lib.check(_lib.load().sea_performer_plan(N, H, T, D, nb, _lib.dtype_code(dtype), nseg, ws), "sea_performer_plan")
        _PLAN_CACHE[key] = (int(nseg[0]), int(ws[0]))
    return _PLAN_CACHE[key]


@_lib.device_guarded
def performer_value(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pos: torch.Tensor,
                    projection: torch.Tensor, want_avg: bool = False, n_segments: int = None):
    """Causal Performer estimator + both concatenations (one launch; two when the rows are cut into segments).
    q,k,v (N,H,T,D); pos (>=T, D) = v_eye_learned_causal[0,0]; projection (nb, D).
    Returns performer_value (N,H,T,3D) = [ctx(pos) | ctx(v) | v] in q's dtype; with want_avg (see
    performer_avg_supported) also the cumulative average of v, (N,H,T,D) -- the `cumavg` of step K.
    n_segments: None = the library's plan for the shape; 1 = the single sequential pass."""
    lib = _lib.load()
    _lib.require_gpu(q, k, v, pos, projection)
    N, H, T, D = q.shape
    assert k.shape == q.shape and v.shape == q.shape and pos.shape[-1] == D and pos.shape[0] >= T
    nb = projection.shape[0]
    q, k, v = (t if t.stride(-1) == 1 else t.contiguous() for t in (q, k, v))
    k = k if k.dtype == q.dtype else k.to(q.dtype)
    v = v if v.dtype == q.dtype else v.to(q.dtype)
    pos = pos if pos.dtype == q.dtype else pos.to(q.dtype)
    pos = pos if pos.stride(-1) == 1 else pos.contiguous()
    # the reference casts the projection buffer to the data dtype before using it
    proj = _cached("proj", (projection,), q.dtype, lambda: projection.to(q.dtype).float().contiguous())
    out = torch.empty((N, H, T, 3 * D), dtype=q.dtype, device=q.device)
    avg = None
    if want_avg:
        assert performer_avg_supported(q, nb)
        avg = torch.empty((N, H, T, D), dtype=q.dtype, device=q.device)
    nseg, ws_bytes = performer_plan(N, H, T, D, nb, q.dtype)
    if n_segments is not None and n_segments != nseg:                 # caller's choice (tests, A/B timing)
        nseg = int(n_segments)
        one_pair = performer_plan(1, 1, 4096, D, nb, q.dtype)         # a shape the plan always cuts: bytes per (pair, segment)
        ws_bytes = N * H * (nseg - 1) * (one_pair[1] // (one_pair[0] - 1))
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=q.device) if nseg > 1 else None
    _lib.check(lib.sea_performer_causal_segmented(
        _p(q), _p(k), _p(v), _p(pos), _lib.dtype_code(q.dtype), _p(proj), N, H, T, D, nb, _lib.strides3(q), _lib.strides3(k),
        _lib.strides3(v), pos.stride(0), _p(out), _p(avg), nseg, _p(ws), ws_bytes, _lib.stream_ptr()),
        "sea_performer_causal_segmented")
    return (out, avg) if want_avg else out


def performer_chunk_rows(D: int, nb: int, dtype) -> int:
    """Rows per chunk of the 16-bit Performer kernel for this shape (`sea_performer_chunk_rows`; 0 = no chunk-aligned step)."""
    return int(_lib.load().sea_performer_chunk_rows(int(D), int(nb), _lib.dtype_code(dtype)))


@_lib.device_guarded
def performer_step(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pos: torch.Tensor, projection: torch.Tensor,
                   state_in: torch.Tensor = None, t_base: int = 0, want_avg: bool = True, n_segments: int = 1,
                   t_base_dev: Optional[torch.Tensor] = None):
    """Stateful, chunk-aligned causal Performer (`sea_performer_causal_step`).  q (N,H,T_new,D) are the NEW rows of sequences
    that have seen `t_base` rows; k, v (N,H,>=t_base+T_new,D) are the kv-cache FROM ROW 0 (the call reads them from the last
    chunk boundary c0 <= t_base on: the open chunk is walked again, which is what makes the rows bitwise the stateless
    pass's); `pos` is the value-embedding table from row 0; `state_in` the image a previous call returned (the state at c0;
    None while no chunk has completed).  Returns (performer_value (N,H,T_new,3D), cumulative average of v for the new rows
    or None, state_out = the image at the last chunk boundary <= t_base + T_new; a new tensor, the input is kept).
    Decode form (`sea_performer_causal_step_at`, a step replayed as a HIP graph): `t_base_dev` is a one-element int32
    device tensor holding the rows seen so far, k / v the fixed-capacity caches (already holding the new row), and
    `state_in` is updated in place."""
    lib = _lib.load()
    _lib.require_gpu(q, k, v, pos, projection)
    N, H, T, D = q.shape
    nb = projection.shape[0]
    assert k.shape[:2] == (N, H) and k.shape[-1] == D and v.shape == k.shape and pos.shape[-1] == D
    C = performer_chunk_rows(D, nb, q.dtype)
    assert C > 0, "the stateful Performer runs on the 16-bit MFMA kernels (bf16 / fp16 data, D in {64, 80, 128})"
    q = q if q.stride(-1) == 1 else q.contiguous()
    k = k if k.dtype == q.dtype else k.to(q.dtype)
    v = v if v.dtype == q.dtype else v.to(q.dtype)
    pos = pos if pos.dtype == q.dtype else pos.to(q.dtype)
    proj = _cached("proj", (projection,), q.dtype, lambda: projection.to(q.dtype).float().contiguous())
    sb = int(lib.sea_performer_state_bytes(N, H, D, nb, _lib.dtype_code(q.dtype)))
    assert sb > 0, "unsupported head size / feature count"
    if t_base_dev is not None:
        assert t_base_dev.dtype == torch.int32 and t_base_dev.numel() == 1 and t_base_dev.is_cuda
        assert state_in is not None and t_base == 0 and n_segments == 1
        kc, vc, pc = k, v, pos                                   # cache / table bases: the kernel finds the boundary
        assert pos.shape[0] >= k.shape[2]
    else:
        c0 = (int(t_base) // C) * C
        assert k.shape[2] >= t_base + T and pos.shape[0] >= t_base + T, "k / v / pos must cover the rows seen plus the new ones"
        assert (state_in is None) == (t_base == 0), "a state image goes with the number of rows it has seen"
        kc, vc, pc = k[:, :, c0:t_base + T], v[:, :, c0:t_base + T], pos[c0:t_base + T]
    kc = kc if kc.stride(-1) == 1 else kc.contiguous()
    vc = vc if vc.stride(-1) == 1 else vc.contiguous()
    pc = pc if pc.stride(-1) == 1 else pc.contiguous()
    assert pc.data_ptr() % 16 == 0 and kc.data_ptr() % 16 == 0 and vc.data_ptr() % 16 == 0, "rows must start 16-byte aligned"
    if state_in is not None:
        assert state_in.dtype == torch.float32 and state_in.numel() * 4 == sb and state_in.is_contiguous()
    state_out = state_in if t_base_dev is not None else torch.empty((sb // 4,), dtype=torch.float32, device=q.device)
    out = torch.empty((N, H, T, 3 * D), dtype=q.dtype, device=q.device)
    avg = None
    if want_avg:
        assert performer_avg_supported(q, nb)
        avg = torch.empty((N, H, T, D), dtype=q.dtype, device=q.device)
    ws, ws_bytes = None, 0
    if n_segments > 1:
        one_pair = performer_plan(1, 1, 4096, D, nb, q.dtype)
        ws_bytes = N * H * (n_segments - 1) * (one_pair[1] // (one_pair[0] - 1))
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=q.device)
    if t_base_dev is not None:
        _lib.check(lib.sea_performer_causal_step_at(
            _p(q), _p(kc), _p(vc), _p(pc), _lib.dtype_code(q.dtype), _p(proj), N, H, T, D, nb, _lib.strides3(q), _lib.strides3(kc),
            _lib.strides3(vc), pc.stride(0), _p(out), _p(avg), _p(state_in), _p(state_out), sb, _p(t_base_dev),
            _lib.stream_ptr()), "sea_performer_causal_step_at")
        return out, avg, state_out
    _lib.check(lib.sea_performer_causal_step(
        _p(q), _p(kc), _p(vc), _p(pc), _lib.dtype_code(q.dtype), _p(proj), N, H, T, D, nb, _lib.strides3(q), _lib.strides3(kc),
        _lib.strides3(vc), pc.stride(0), _p(out), _p(avg), _p(state_in), _p(state_out), sb, int(t_base), int(n_segments),
        _p(ws), ws_bytes, _lib.stream_ptr()), "sea_performer_causal_step")
    return out, avg, state_out


def predictor_mlp_supported(D1: int, D2: int, H: int, Din: int) -> bool:
    """Shapes csrc/sea_mlp.hip is instantiated for (launch_mlp)."""
    if (D1, D2) == (256, 128):                                    # d = 128: encoder weights streamed through LDS
        return H % 4 == 0 and Din % 8 == 0 and Din <= 384
    if D1 == 128:                                                 # d = 64: every predictor length T_M = 2 * D2, T_M % 32 == 0, <= 512
        return D2 % 16 == 0 and 16 <= D2 <= 256 and H % 4 == 0 and Din % 8 == 0 and Din <= 256
    return (D1, D2) == (160, 128) and H % 4 == 0 and Din % 8 == 0 and Din <= 256


def _pack_a_fragments(w: torch.Tensor, kperm=None) -> torch.Tensor:
    """(M, K) weight -> MFMA 16x16x32 A fragments [K/32][M/16][64 lanes][8]: lane l holds W[16*tile + l%16][k(ks, l//16, j)],
    k = 32*ks + 8*g + j by default, or kperm[ks, g, j].  M, K already padded to multiples of 16 / 32."""
    M, K = w.shape
    ks = torch.arange(K // 32, device=w.device).view(-1, 1, 1, 1, 1)
    tile = torch.arange(M // 16, device=w.device).view(1, -1, 1, 1, 1)
    g = torch.arange(4, device=w.device).view(1, 1, -1, 1, 1)
    li = torch.arange(16, device=w.device).view(1, 1, 1, -1, 1)
    j = torch.arange(8, device=w.device).view(1, 1, 1, 1, -1)
    k = (32 * ks + 8 * g + j) if kperm is None else kperm(ks, g, j)
    return w[(16 * tile + li).expand(K // 32, M // 16, 4, 16, 8), k.expand(K // 32, M // 16, 4, 16, 8)].contiguous()


@_lib.device_guarded
def predictor_mlp(x: torch.Tensor, enc_lin, enc_ln, dec_lin, ln1, scaler_lin, want_tpred: bool = False,
                  x_c8_out: Optional[torch.Tensor] = None):
    """Fused predictor MLP (csrc/sea_mlp.hip): x (N,H,T,Din) -> (x_c8 (N,T,H*2/8,Wd,8), t_pred or None,
    row_scale (N,H,T) fp32, avg_scale (N,H,T) fp32).  enc_lin/dec_lin/scaler_lin: nn.Linear; enc_ln/ln1: nn.LayerNorm.
    `x_c8_out`: a preallocated (N,T,H*2/8,Wd,8) VIEW whose batch items may lie apart (rows of an item dense): a decode
    session lets the new row land behind each item's CNN window."""
    lib = _lib.load()
    _lib.require_gpu(x)
    N, H, T, Din = x.shape
    dt = x.dtype
    D1, D2 = enc_lin.out_features, dec_lin.out_features
    Wd = D2 // 2
    assert enc_lin.in_features == Din and dec_lin.in_features == D1 and scaler_lin.in_features == D1 and scaler_lin.out_features == 2
    assert predictor_mlp_supported(D1, D2, H, Din) and tuple(ln1.normalized_shape) == (Wd,)
    if x.stride(-1) != 1:
        x = x.contiguous()

    def build():
        KP = (Din + 31) // 32 * 32
        w1 = torch.zeros((D1, KP), dtype=dt, device=x.device)
        w1[:, :Din] = enc_lin.weight.to(dt)
        # each half (split) of the decoder owns ceil(Wd / 16) whole 16-row tiles; rows past Wd are zero (T_M = 96: Wd = 24)
        WdP = (Wd + 15) // 16 * 16
        w2 = torch.zeros((2 * WdP + 16, D1), dtype=dt, device=x.device)
        w2[:Wd] = dec_lin.weight[:Wd].to(dt)
        w2[WdP:WdP + Wd] = dec_lin.weight[Wd:].to(dt)
        w2[2 * WdP:2 * WdP + 2] = scaler_lin.weight.to(dt)
        w1p = _pack_a_fragments(w1)
        w2p = _pack_a_fragments(w2, kperm=lambda ks, g, j: 16 * (2 * ks + j // 4) + 4 * g + j % 4)
        f = lambda t: t.to(dt).float().reshape(-1)

        def padded(t, parts):                                     # `parts` runs of Wd values, each zero-padded to WdP
            o = torch.zeros((parts, WdP), dtype=torch.float32, device=x.device)
            o[:, :Wd] = f(t).view(parts, Wd)
            return o.reshape(-1)
        vec = torch.cat([f(enc_lin.bias), f(enc_ln.weight), f(enc_ln.bias), padded(dec_lin.bias, 2), padded(ln1.weight, 1),
                         padded(ln1.bias, 1), f(scaler_lin.bias)]).contiguous()
        return w1p, w2p, vec
    w1p, w2p, vec = _cached("mlp", (enc_lin.weight, enc_lin.bias, enc_ln.weight, enc_ln.bias, dec_lin.weight, dec_lin.bias,
                                    ln1.weight, ln1.bias, scaler_lin.weight, scaler_lin.bias), dt, build)
    if x_c8_out is None:
        x_c8, xs_n = torch.empty((N, T, H * 2 // 8, Wd, 8), dtype=dt, device=x.device), 0
    else:
        x_c8, xs_n = x_c8_out, x_c8_out.stride(0)
        assert tuple(x_c8.shape) == (N, T, H * 2 // 8, Wd, 8) and x_c8.dtype == dt and x_c8.stride()[1:] == (H * 2 // 8 * Wd * 8, Wd * 8, 8, 1)
    tpred = torch.empty((N, H, T, D1), dtype=dt, device=x.device) if want_tpred else None
    row_scale = torch.empty((N, H, T), dtype=torch.float32, device=x.device)
    avg_scale = torch.empty((N, H, T), dtype=torch.float32, device=x.device)
    _lib.check(lib.sea_predictor_mlp(_p(x), _lib.dtype_code(dt), N, H, T, Din, _lib.strides3(x), D1, D2, _p(w1p), _p(w2p),
                                     _p(vec), float(enc_ln.eps), float(ln1.eps), _p(x_c8), int(xs_n), _p(tpred), _p(row_scale),
                                     _p(avg_scale), _lib.stream_ptr()), "sea_predictor_mlp")
    return x_c8, tpred, row_scale, avg_scale


def to_c8(x: torch.Tensor) -> torch.Tensor:
    """(N, C, T, W) -> the channel-blocked layout of the conv kernels, a dense 5-D tensor (N, T, C/8, W, 8)."""
    N, C, T, W = x.shape
    assert C % 8 == 0
    return x.reshape(N, C // 8, 8, T, W).permute(0, 3, 1, 4, 2).contiguous()


def from_c8(y: torch.Tensor) -> torch.Tensor:
    """Inverse of to_c8: (N, T, C/8, W, 8) -> contiguous (N, C, T, W)."""
    N, T, C8, W, _e = y.shape
    return y.permute(0, 2, 4, 1, 3).reshape(N, C8 * 8, T, W).contiguous()


@_lib.device_guarded
def split_layernorm_c8(x: torch.Tensor, splits: int, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5):
    """ChannelSplit + LayerNorm writing the C8 layout: (N, C, T, splits*W) -> (N, T, C*splits/8, W, 8).  16-bit and fp32 data."""
    lib = _lib.load()
    _lib.require_gpu(x, weight, bias)
    N, C, T, SW = x.shape
    W = SW // splits
    x = x.contiguous()
    w, b = _cached("ln", (weight, bias), x.dtype, lambda: (weight.to(x.dtype).contiguous(), bias.to(x.dtype).contiguous()))
    out = torch.empty((N, T, C * splits // 8, W, 8), dtype=x.dtype, device=x.device)
    _lib.check(lib.sea_split_layernorm_c8(_p(x), _lib.dtype_code(x.dtype), N, C, T, splits, W, _p(w), _p(b), float(eps),
                                          _p(out), _lib.stream_ptr()), "sea_split_layernorm_c8")
    return out


def pack_conv_weight(weight: torch.Tensor, ksize: int, dtype: torch.dtype):
    """(Cout, Cin, >=ksize, ksize) -> (Cout, ksize*ksize*CinP) laid out [co][tap][ci], ci zero-padded to a multiple
    of 32 (16-bit kernels: one MFMA k-step) or 16 (the fp32 kernel: four).  Only the first `ksize` kernel rows are live for
    the causal conv (modules.py:113-121)."""
    Cout, Cin = weight.shape[:2]
    pad = 16 if dtype == torch.float32 else 32
    CinP = (Cin + pad - 1) // pad * pad
    w = weight[:, :, :ksize, :].to(dtype).permute(0, 2, 3, 1)                  # (Cout, k, k, Cin)
    packed = torch.zeros((Cout, ksize, ksize, CinP), dtype=dtype, device=weight.device)
    packed[..., :Cin] = w
    return packed.reshape(Cout, ksize * ksize * CinP).contiguous(), CinP


@_lib.device_guarded
def causal_conv_c8(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, ksize: int, dilation: int, pad_w: int,
                   relu: bool = True) -> torch.Tensor:
    """Causal (along T) dilated conv + bias (+ReLU) on a C8 activation (N, T, Cin/8, W, 8).
    `weight` is the module's (Cout, Cin, 2k-1, k) parameter.  Returns (N, T, Cout/8, W, 8).  16-bit data: bf16 / f16 MFMA
    (`sea_causal_conv_c8`); fp32 data: the fp32 MFMA, exact products (`sea_causal_conv_c8_f32`, `conv_c8_f32_supported`)."""
    lib = _lib.load()
    _lib.require_gpu(x, weight, bias)
    N, T, C8, W, _e = x.shape
    assert _e == 8 and x.is_contiguous(), "input must be a dense C8 activation (N, T, Cin/8, W, 8)"
    Cin, Cout = C8 * 8, weight.shape[0]
    assert weight.shape[1] == Cin and Cout % 8 == 0
    (wp, CinP), bf = _cached("conv", (weight, bias), x.dtype,
                             lambda: (pack_conv_weight(weight, ksize, x.dtype), bias.to(x.dtype).float().contiguous()))
    y = torch.empty((N, T, Cout // 8, W, 8), dtype=x.dtype, device=x.device)
    if x.dtype == torch.float32:
        assert conv_c8_f32_supported(Cin, Cout, ksize)
        _lib.check(lib.sea_causal_conv_c8_f32(_p(x), N, T, W, Cin, Cout, _p(wp), CinP, _p(bf), int(ksize), int(dilation),
                                              int(pad_w), int(relu), _p(y), _lib.stream_ptr()), "sea_causal_conv_c8_f32")
        return y
    _lib.check(lib.sea_causal_conv_c8(_p(x), _lib.dtype_code(x.dtype), N, T, W, Cin, Cout, _p(wp), CinP, _p(bf),
                                      int(ksize), int(dilation), int(pad_w), int(relu), _p(y), _lib.stream_ptr()),
               "sea_causal_conv_c8")
    return y


def conv_c8_f32_supported(Cin: int, Cout: int, ksize: int) -> bool:
    """Shapes `sea_causal_conv_c8_f32` takes: the fp32 weight image must fit the 160 KB LDS (24 -> 24 channels: 37 KB,
    64 -> 64: 148 KB; 80 -> 80 does not)."""
    nt, cinp = (Cout + 15) // 16, (Cin + 15) // 16 * 16
    return (ksize in (1, 3) and Cin % 8 == 0 and Cout % 8 == 0 and Cout <= 80
            and 16 * nt * (ksize * ksize * cinp * 4 + 4) <= 160 * 1024)


def conv_z_supported(Cout: int, H: int, ksize: int, W: int) -> bool:
    """Shapes `sea_causal_conv_c8_z` takes: a 3 x 3 layer of at most 80 output channels (5 MFMA tiles), H <= 64."""
    return ksize == 3 and Cout % 8 == 0 and Cout <= 80 and 0 < H <= 64 and W % 4 == 0


@_lib.device_guarded
def causal_conv_c8_z(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, ksize: int, dilation: int, pad_w: int,
                     conv1x1_w: torch.Tensor, conv1x1_b: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor,
                     relu: bool = True, want_y: bool = False):
    """`causal_conv_c8` of the LAST (conv, ReLU) pair of the predictor CNN with the tail's 1x1 convolution in its epilogue
    (`sea_causal_conv_c8_z`).  conv1x1_w (H, Cout) = the live row of the 1x1 kernel, conv1x1_b (H); ln_w / ln_b are only the
    rest of the tail's cache key (one pack serves the conv epilogue and the tail launches).
    Returns (y or None, z): z (N, T, H, W) fp32 = W1 . relu(conv(x) + bias) + b1 -- bit for bit what the tail kernels compute
    from y; y (N, T, Cout/8, W, 8) only with `want_y`."""
    lib = _lib.load()
    _lib.require_gpu(x, weight, bias, conv1x1_w, conv1x1_b)
    N, T, C8, W, _e = x.shape
    assert _e == 8 and x.is_contiguous(), "input must be a dense C8 activation (N, T, Cin/8, W, 8)"
    Cin, Cout, H = C8 * 8, weight.shape[0], conv1x1_w.shape[0]
    assert weight.shape[1] == Cin and conv1x1_w.shape == (H, Cout) and conv_z_supported(Cout, H, ksize, W)
    (wp, CinP), bf = _cached("conv", (weight, bias), x.dtype,
                             lambda: (pack_conv_weight(weight, ksize, x.dtype), bias.to(x.dtype).float().contiguous()))
    _cw, cb, _g, _b, w16, Cp = _tail_pack(conv1x1_w, conv1x1_b, ln_w, ln_b, x.dtype, x.device)
    y = torch.empty((N, T, Cout // 8, W, 8), dtype=x.dtype, device=x.device) if want_y else None
    z = torch.empty((N, T, H, W), dtype=torch.float32, device=x.device)
    _lib.check(lib.sea_causal_conv_c8_z(_p(x), _lib.dtype_code(x.dtype), N, T, W, Cin, Cout, _p(wp), CinP, _p(bf),
                                        int(ksize), int(dilation), int(pad_w), int(relu), _p(y), _p(w16), Cp, _p(cb), H, _p(z),
                                        _lib.stream_ptr()), "sea_causal_conv_c8_z")
    return y, z


@_lib.device_guarded
def decode_stage(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_in: torch.Tensor, kv_cache: torch.Tensor,
                 counters: torch.Tensor) -> None:
    """`sea_decode_stage`: the new rows of a decoding step, (N,H,1,D) each (any [n,h] strides), into the session's static
    buffers -- q into q_in (N,H,1,D), k / v into kv_cache (2,N,H,capacity,D) at the row the device counters name
    (int32 [seen, tsrc]: row = counters[0])."""
    lib = _lib.load()
    _lib.require_gpu(q, k, v, q_in, kv_cache, counters)
    N, H, one, D = q.shape
    assert one == 1 and k.shape == q.shape and v.shape == q.shape and q.dtype == k.dtype == v.dtype == q_in.dtype == kv_cache.dtype
    assert q_in.is_contiguous() and tuple(q_in.shape) == (N, H, 1, D) and kv_cache.is_contiguous() and kv_cache.shape[:3] == (2, N, H)
    assert counters.dtype == torch.int32 and counters.numel() == 2 and counters.is_contiguous()
    # the kernel moves whole 16-byte vectors: rows contiguous along D, [n, h] strides in multiples of 8 elements, 16-byte aligned
    # base.  Anything else (e.g. rows sliced out of a fused qkv buffer at an odd offset) is repacked here -- the session used
    # to take any layout through `.copy_`
    ok = lambda t: t.stride(-1) == 1 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0
    q, k, v = (t if ok(t) else t.contiguous() for t in (q, k, v))
    st = lambda t: (ctypes_i64 * 2)(t.stride(0), t.stride(1))
    _lib.check(lib.sea_decode_stage(_p(q), _p(k), _p(v), _lib.dtype_code(q.dtype), N, H, D, st(q), st(k), st(v), _p(q_in),
                                    _p(kv_cache), kv_cache.shape[3], _p(counters), _lib.stream_ptr()), "sea_decode_stage")


def decode_cnn_supported(C: int, H: int, T_m: int, dtype) -> bool:
    """Shapes `sea_decode_cnn_tail_select` is instantiated for (csrc/sea_topk.hip: launch_decode_cnn)."""
    return (dtype in (torch.float16, torch.bfloat16) and T_m == 256 and H % 4 == 0 and 0 < H <= 40 and C == 2 * H and C % 8 == 0)


def decode_cnn_emits(C: int) -> bool:
    """Does `sea_decode_cnn_tail_select` expand the CSR columns itself for this channel count?"""
    return C <= 64


@_lib.device_guarded
def decode_cnn_tail_select(x_new: torch.Tensor, x_ring: torch.Tensor, y1_ring: torch.Tensor, y2: torch.Tensor, conv1, conv2,
                           conv_w: torch.Tensor, conv_b: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, T_m: int,
                           keep: torch.Tensor, k: int, counters: torch.Tensor, ticket: torch.Tensor, crow_out: torch.Tensor,
                           is_causal: bool = True, eps: float = 1e-5, want_probs: bool = True,
                           col_out: Optional[torch.Tensor] = None, T_cap: int = 0):
    """One launch for a decoding step's predictor CNN (two 3 x 3 dilated causal convolutions, one new row each), tail,
    selection and the advance of the session's device counters (`sea_decode_cnn_tail_select`).  `conv1` / `conv2` are the
    `CausalConv2d` modules; rings and counters as include/sea_hip.h describes.  `col_out` (N, z_cap) int32 with `T_cap` (C <=
    64, `decode_cnn_emits`): the CSR columns of the new row are written by this launch too.  Returns (probs or None, (bits,
    row_nnz, head_off))."""
    lib = _lib.load()
    _lib.require_gpu(x_new, x_ring, y1_ring, y2, conv_w, conv_b, ln_w, ln_b, keep, counters, ticket, crow_out)
    N, C8, W4, _e = x_new.shape[0], x_new.shape[-3], x_new.shape[-2], x_new.shape[-1]
    C, H, dt, dev = C8 * 8, conv_w.shape[0], x_new.dtype, x_new.device
    assert _e == 8 and x_new.is_contiguous() and x_ring.is_contiguous() and y1_ring.is_contiguous() and y2.is_contiguous()
    assert x_ring.shape[0] == N and tuple(x_ring.shape[2:]) == (C8, W4, 8) and tuple(y1_ring.shape[2:]) == (C8, W4, 8)
    assert decode_cnn_supported(C, H, T_m, dt) and W4 * 4 == T_m
    for cv in (conv1, conv2):
        assert cv.kernel_size == 3 and cv.in_channels == C and cv.out_channels == C and cv.padding[1] == cv.dilation == conv1.dilation
    assert counters.dtype == torch.int32 and counters.numel() == 3 and ticket.dtype == torch.int32 and ticket.numel() == 1
    assert keep.dtype == torch.int32 and keep.ndim == 1 and crow_out.dtype == torch.int32 and tuple(crow_out.shape) == (N, 2)
    packs = []
    for cv in (conv1, conv2):
        packs.append(_cached("conv", (cv.weight, cv.bias), dt,
                             lambda cv=cv: (pack_conv_weight(cv.weight, 3, dt), cv.bias.to(dt).float().contiguous())))
    (w1p, CinP), b1 = packs[0]
    (w2p, _c), b2 = packs[1]
    _cw, cb, g, b, w16, Cp = _tail_pack(conv_w, conv_b, ln_w, ln_b, dt, dev)
    tab = _tail_consts(ln_w, ln_b, 4, T_m, dt, dev)
    probs = torch.empty((N, H, 1, T_m), dtype=dt, device=dev) if want_probs else None
    Wb = (H * T_m + 31) // 32
    bits = torch.empty((N, 1, Wb), dtype=torch.int32, device=dev)
    row_nnz = torch.empty((N, 1), dtype=torch.int32, device=dev)
    head_off = torch.empty((N, 1, H + 1), dtype=torch.int32, device=dev)
    _lib.check(lib.sea_decode_cnn_tail_select(
        _p(x_new), _p(x_ring), _p(y1_ring), _p(y2), _lib.dtype_code(dt), N, C, H, W4, x_ring.shape[1], y1_ring.shape[1],
        _p(w1p), _p(b1), _p(w2p), _p(b2), CinP, int(conv1.dilation), int(conv1.padding[1]), _p(cb), _p(w16), Cp, _p(g), _p(b),
        float(eps), _p(probs), _p(keep), _p(counters), _p(ticket), int(is_causal), int(k), _p(bits), _p(row_nnz), _p(head_off),
        _p(crow_out), _p(col_out), col_out.stride(0) if col_out is not None else 0, col_out.shape[1] if col_out is not None else 0,
        int(T_cap), _p(tab), _lib.stream_ptr()), "sea_decode_cnn_tail_select")
    return probs, (bits, row_nnz, head_off)


@_lib.device_guarded
def c8_window_shift(xs: torch.Tensor, counters: Optional[torch.Tensor] = None) -> None:
    """`sea_c8_window_shift`: xs (N, rows, ...) dense -> xs[:, r] = xs[:, r + 1] in place (the last row keeps its values);
    `counters`: two device int32 the same launch advances by one (the last launch of a decoding step)."""
    lib = _lib.load()
    _lib.require_gpu(xs)
    assert xs.is_contiguous() and xs.dim() >= 3
    N, rows = xs.shape[:2]
    row_bytes = xs[0, 0].numel() * xs.element_size()
    assert counters is None or (counters.dtype == torch.int32 and counters.numel() == 2 and counters.is_contiguous())
    _lib.check(lib.sea_c8_window_shift(_p(xs), N, rows, row_bytes, _p(counters), _lib.stream_ptr()), "sea_c8_window_shift")
