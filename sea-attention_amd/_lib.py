"""ctypes binding of libsea_hip.so (include/sea_hip.h).  The HIP library IS the product path:
if it is missing this module raises -- there is no CPU or eager fallback."""
import ctypes
import os
from ctypes import c_int, c_int32, c_int64, c_void_p, c_char_p, POINTER

import torch

from . import _build

SEA_F32, SEA_F16, SEA_BF16 = 0, 1, 2
ABI_VERSION = 3            # include/sea_hip.h: SEA_ABI_VERSION
_DTYPES = {torch.float32: SEA_F32, torch.float16: SEA_F16, torch.bfloat16: SEA_BF16}

_lib = None

i64 = c_int64
ptr = c_void_p
_i64p = POINTER(c_int64)

_SIGNATURES = {
    "sea_version": ([], c_int),
    "sea_last_error": ([], c_char_p),
    "sea_topk_select": ([ptr, c_int, i64, i64, i64, i64, i64, i64, i64, ptr, i64, i64, c_int, c_int,
                         ptr, ptr, ptr, ptr, ptr], c_int),
    "sea_mask_to_bits": ([ptr, c_int, i64, i64, i64, i64, i64, i64, i64, i64, c_int, c_int,
                          ptr, ptr, ptr, ptr], c_int),
    "sea_csr_row_scan": ([ptr, i64, i64, ptr, c_int, ptr], c_int),
    "sea_csr_emit": ([ptr, ptr, ptr, i64, i64, i64, i64, i64, c_int, c_int, ptr, c_int, i64, i64, ptr, ptr], c_int),
    "sea_csr_head_offsets": ([ptr, ptr, c_int, i64, i64, i64, i64, i64, ptr, ptr], c_int),
    "sea_csr_sddmm": ([ptr, ptr, c_int, i64, i64, i64, i64, i64, _i64p, _i64p, ptr, ptr, c_int, i64, ptr, ptr], c_int),
    "sea_csr_softmax": ([ptr, ptr, i64, i64, i64, i64, ptr, ptr, c_int, i64, ptr], c_int),
    "sea_csr_elmul": ([ptr, ptr, ptr, c_int, _i64p, i64, i64, i64, i64, ptr, ptr, c_int, i64, ptr], c_int),
    "sea_csr_spmm": ([ptr, ptr, c_int, i64, i64, i64, i64, i64, _i64p, ptr, ptr, c_int, i64, ptr, ptr, ptr], c_int),
    "sea_sparse_attention": ([ptr, ptr, ptr, c_int, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p,
                              ptr, ptr, i64, ptr, ptr, ptr, _i64p, ptr, ptr, c_int, _i64p, ptr], c_int),
    "sea_sparse_attention_ex": ([ptr, ptr, ptr, c_int, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p,
                                 ptr, ptr, i64, ptr, ptr, ptr, _i64p, ptr, ptr, c_int, _i64p, ptr, i64, ptr, c_int, ptr], c_int),
    "sea_sparse_attention_fused": ([ptr, ptr, ptr, c_int, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p,
                                    ptr, ptr, i64, ptr, ptr, ptr, _i64p, ptr, ptr, c_int, _i64p, ptr, i64, ptr, i64, c_int, c_int, c_int, ptr], c_int),
    "sea_sparse_attention_fused_at": ([ptr, ptr, ptr, c_int, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p,
                                       ptr, ptr, i64, ptr, ptr, ptr, _i64p, ptr, ptr, c_int, _i64p, ptr, i64, ptr, c_int, c_int, c_int, ptr], c_int),
    "sea_sparse_attention_bwd": ([ptr, ptr, ptr, c_int, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p, ptr, ptr, i64, ptr,
                                  ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr], c_int),
    "sea_sparse_attention_bwd_workspace_bytes": ([i64, i64, i64, i64], i64),
    "sea_sparse_attention_bwd_gather": ([ptr, ptr, ptr, c_int, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p, ptr, ptr, i64, ptr,
                                         ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, i64, ptr], c_int),
    "sea_attention_plan": ([ptr, i64, i64, i64, i64, i64, c_int, ctypes.c_float, ptr, ptr], c_int),
    "sea_sparse_attention_bytes": ([i64, i64, i64, i64, i64, c_int], i64),
    "sea_split_layernorm": ([ptr, c_int, i64, i64, i64, i64, i64, ptr, ptr, ctypes.c_float, c_int, ptr, ptr], c_int),
    "sea_predictor_tail": ([ptr, c_int, i64, i64, i64, i64, i64, i64, i64, _i64p, ptr, ptr, ptr, i64, ptr, ptr,
                            ctypes.c_float, ptr, ptr, ptr], c_int),
    "sea_cumavg": ([ptr, c_int, i64, i64, i64, i64, _i64p, ptr, ptr], c_int),
    "sea_cumavg_sliced": ([ptr, c_int, i64, i64, i64, i64, _i64p, ptr, i64, ptr, i64, ptr], c_int),
    "sea_predictor_tail_select": ([ptr, c_int, i64, i64, i64, i64, i64, i64, i64, _i64p, ptr, ptr, i64, ptr, ptr,
                                   ctypes.c_float, ptr, ptr, ptr, i64, i64, c_int, c_int, ptr, ptr, ptr, ptr, ptr], c_int),
    "sea_predictor_tail_consts": ([c_int, i64, i64, i64, ptr, ptr, ptr, ptr], c_int),
    "sea_predictor_mlp": ([ptr, c_int, i64, i64, i64, i64, _i64p, i64, i64, ptr, ptr, ptr, ctypes.c_float,
                           ctypes.c_float, ptr, i64, ptr, ptr, ptr, ptr], c_int),
    "sea_split_layernorm_c8": ([ptr, c_int, i64, i64, i64, i64, i64, ptr, ptr, ctypes.c_float, ptr, ptr], c_int),
    "sea_causal_conv_c8": ([ptr, c_int, i64, i64, i64, i64, i64, ptr, i64, ptr, c_int, c_int, c_int, c_int, ptr, ptr], c_int),
    "sea_decode_cnn_tail_select": ([ptr, ptr, ptr, ptr, c_int, i64, i64, i64, i64, i64, i64, ptr, ptr, ptr, ptr, i64, c_int, c_int,
                                    ptr, ptr, i64, ptr, ptr, ctypes.c_float, ptr, ptr, ptr, ptr, c_int, c_int, ptr, ptr, ptr, ptr,
                                    ptr, i64, i64, i64, ptr, ptr], c_int),
    "sea_causal_conv_c8_f32": ([ptr, i64, i64, i64, i64, i64, ptr, i64, ptr, c_int, c_int, c_int, c_int, ptr, ptr], c_int),
    "sea_causal_conv_c8_z": ([ptr, c_int, i64, i64, i64, i64, i64, ptr, i64, ptr, c_int, c_int, c_int, c_int, ptr,
                              ptr, i64, ptr, i64, ptr, ptr], c_int),
    "sea_predictor_tail_z": ([ptr, c_int, i64, i64, i64, i64, i64, i64, ptr, ptr, ptr, ctypes.c_float, ptr, ptr, ptr], c_int),
    "sea_predictor_tail_select_z": ([ptr, c_int, i64, i64, i64, i64, i64, i64, ptr, ptr, ptr, ctypes.c_float, ptr, ptr,
                                     ptr, i64, i64, c_int, c_int, ptr, ptr, ptr, ptr, ptr], c_int),
    "sea_performer_causal": ([ptr, ptr, ptr, ptr, c_int, ptr, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p, i64, ptr, ptr, ptr], c_int),
    "sea_performer_state_bytes": ([i64, i64, i64, i64, c_int], i64),
    "sea_performer_chunk_rows": ([i64, i64, c_int], i64),
    "sea_attention_few_rows": ([], i64),
    "sea_performer_causal_step": ([ptr, ptr, ptr, ptr, c_int, ptr, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p, i64, ptr, ptr,
                                   ptr, ptr, i64, i64, i64, ptr, i64, ptr], c_int),
    "sea_performer_causal_step_at": ([ptr, ptr, ptr, ptr, c_int, ptr, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p, i64, ptr, ptr,
                                      ptr, ptr, i64, ptr, ptr], c_int),
    "sea_predictor_tail_select_at": ([ptr, c_int, i64, i64, i64, i64, i64, i64, i64, _i64p, ptr, ptr, i64, ptr, ptr,
                                      ctypes.c_float, ptr, ptr, ptr, ptr, c_int, c_int, ptr, ptr, ptr, ptr, ptr, ptr], c_int),
    "sea_decode_stage": ([ptr, ptr, ptr, c_int, i64, i64, i64, _i64p, _i64p, _i64p, ptr, ptr, i64, ptr, ptr], c_int),
    "sea_c8_window_shift": ([ptr, i64, i64, i64, ptr, ptr], c_int),
    "sea_csr_emit_at": ([ptr, ptr, i64, i64, i64, i64, ptr, i64, c_int, c_int, ptr, c_int, i64, i64, ptr], c_int),
    "sea_performer_avg_supported": ([i64, i64, c_int], c_int),
    "sea_performer_plan": ([i64, i64, i64, i64, i64, c_int, _i64p, _i64p], c_int),
    "sea_performer_causal_segmented": ([ptr, ptr, ptr, ptr, c_int, ptr, i64, i64, i64, i64, i64, _i64p, _i64p, _i64p, i64, ptr, ptr,
                                        i64, ptr, i64, ptr], c_int),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path():
    return _build.LIB_PATH


SEA_ATTN_AUTO, SEA_ATTN_GATHER, SEA_ATTN_TILE = 0, 1, 2
SEA_OK, SEA_EINVAL, SEA_EUNSUPPORTED = 0, -1, -2


def load(build_if_missing=True):
    """Load libsea_hip.so, (re)building it first when it is missing or was built from other sources than the tree
    holds now (content hash, `_build.is_stale`).  `SEA_HIP_LIB=<path>` loads that file as is (A/B builds of the same
    ABI) and fails if it does not exist.  Raises if the library cannot be had: there is no other path."""
    global _lib
    if _lib is not None:
        return _lib
    override = os.environ.get("SEA_HIP_LIB")
    if override:
        if not os.path.exists(override):
            raise RuntimeError(f"SEA_HIP_LIB={override} does not exist")
        path = override
    else:
        path = _build.LIB_PATH
        if _build.is_stale():
            if not build_if_missing:
                raise RuntimeError(f"{path} is missing or stale; run `python -c 'import __graft_entry__ as g; g.build()'`")
            _build.ensure_built()                  # one builder per tree: the other ranks wait on the lock
    lib = ctypes.CDLL(path)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.sea_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.sea_version()}, this binding speaks {ABI_VERSION} (include/sea_hip.h: "
                           "SEA_ABI_VERSION lists what changed) -- rebuild the library from this tree's csrc/")
    _lib = lib
    return lib


def dtype_code(dt):
    try:
        return _DTYPES[dt]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}; the HIP kernels take float32 / float16 / bfloat16")


def strides3(t):
    """element strides [n, h, t] of a (N,H,T,D) tensor whose last stride is 1, as a C int64[3]."""
    assert t.stride(-1) == 1, "innermost stride must be 1"
    return (c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def strides4(t):
    return (c_int64 * 4)(*t.stride())


def strides5_blocked(t):
    """{n, c, t, w, block-of-8} element strides of an activation for sea_predictor_tail: `t` is either a 4-D
    (N,C,T,W) tensor of any strides or a 5-D C8 tensor (N, T, C/8, W, 8)."""
    if t.dim() == 5:
        sn, st, sb, sw, sc = t.stride()
        return (c_int64 * 5)(sn, sc, st, sw, sb)
    sn, sc, st, sw = t.stride()
    return (c_int64 * 5)(sn, sc, st, sw, 8 * sc)


def stream_ptr():
    """hipStream_t of the CURRENT device's current stream (operators run under `device_guarded`, which makes the
    tensors' device current first)."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def device_guarded(fn):
    """Operator decorator: run `fn` with the device of its first device-tensor argument current, so that the stream
    handed to the library and the launch itself belong to the tensors' device (a tensor on cuda:1 called while cuda:0
    is current would otherwise be launched on cuda:0's stream).  No cost beyond an argument scan when it already is."""
    import functools

    def _first_cuda(args, kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor):
                if a.is_cuda:
                    return a
            else:
                t = getattr(a, "crow", None)              # FlatCSR handle
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    return t
        return None

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t = _first_cuda(args, kwargs)
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(t.device):
            return fn(*args, **kwargs)
    return wrapper


def check(rc, what):
    if rc != 0:
        msg = load().sea_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and type(t).__name__ == "LazyTensor":
            t.materialize()          # what reaches the C ABI is the real tensor (ops.LazyTensor.data_ptr serves it)
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "sea_attention_amd operators run only on an MI355X device tensor through libsea_hip.so; "
                "got a CPU tensor (there is deliberately no CPU fallback)")
