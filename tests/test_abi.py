"""The C-ABI library builds for gfx950, loads, and exports every symbol include/sea_hip.h declares.
No compute call is made here (no GPU in the CPU suite)."""
import ctypes
import os
import re

import pytest
import torch

import sea_attention_amd as S
from sea_attention_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sea_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sea_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_bound():
    decl = _declared_symbols()
    assert len(decl) >= 12
    assert sorted(_lib.EXPORTED_SYMBOLS) == decl


def test_library_builds_and_exports_everything():
    path = _build.build_library()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} not exported"
    lib.sea_version.restype = ctypes.c_int
    assert lib.sea_version() == 3
    # host-only helper: algorithmic bytes, SURVEY 8d (cfg 3, bf16, Z = 8.32 M -> 2.20 GB)
    lib.sea_sparse_attention_bytes.restype = ctypes.c_int64
    lib.sea_sparse_attention_bytes.argtypes = [ctypes.c_int64] * 5 + [ctypes.c_int]
    b = lib.sea_sparse_attention_bytes(8_320_000, 1, 32, 4096, 64, 2)
    assert b == 8_320_000 * (2 * 64 * 2 + 4) + 32 * 4096 * (2 * 64 * 2 + 4)
    assert abs(b / 1e9 - 2.20) < 0.02


def test_bad_arguments_return_error_codes_not_crashes():
    lib = _lib.load()
    rc = lib.sea_csr_row_scan(None, 1, 4, None, 4, None)
    assert rc == -1
    assert b"null pointer" in lib.sea_last_error()
    rc = lib.sea_topk_select(None, 0, 1, 1, 1, 4, 0, 0, 0, None, 0, 1, 1, 1, None, None, None, None, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    """The product path has no CPU fallback: operators raise on CPU input."""
    from sea_attention_amd.perlin_attention import ops
    probs = torch.softmax(torch.randn(1, 2, 8, 8), -1)
    keep = torch.ones(8, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.topk_to_csr(probs, keep, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.resize_from_m_to_t_csr(torch.ones(1, 2, 8, 8), 0, 4)
    crow = torch.tensor([[0, 1, 2]]); col = torch.tensor([[0, 1]])
    csr = torch.sparse_csr_tensor(crow, col, torch.ones(1, 2), size=(1, 2, 4))
    q = torch.randn(1, 1, 2, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.flat_csr_masked_bmm(q, torch.randn(1, 1, 4, 8), csr)


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/sea_hip.h against the ctypes table of _lib.py: same number of parameters, and pointer
    / integer / float kinds line up (a drifted binding would pass garbage without any error)."""
    src = open(os.path.join(ROOT, "include", "sea_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = re.findall(r"\b(?:int|int64_t|const char\s*\*)\s+(sea_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)
    assert len(protos) == len(_lib.EXPORTED_SYMBOLS)

    def kind_of_c(param):
        p = " ".join(param.split())
        if p in ("void", ""):
            return None
        if "*" in p or p.startswith("sea_stream_t"):
            return "ptr"
        if p.startswith("float"):
            return "float"
        return "int"

    def kind_of_ctypes(t):
        if t in (ctypes.c_float, ctypes.c_double):
            return "float"
        if t in (ctypes.c_int, ctypes.c_int32, ctypes.c_int64):
            return "int"
        return "ptr"                                    # c_void_p, c_char_p, POINTER(...)

    for name, params in protos:
        want = [k for k in (kind_of_c(x) for x in params.split(",")) if k is not None]
        argtypes, _res = _lib._SIGNATURES[name]
        got = [kind_of_ctypes(t) for t in argtypes]
        assert got == want, f"{name}: header {want} vs ctypes {got}"


def test_host_side_planning_entries():
    """`sea_performer_plan`, `sea_performer_state_bytes`, `sea_sparse_attention_bytes` are host arithmetic (no device
    work): the plan cuts the rows only when N*H leaves the chip idle, segments are whole 64-row chunks, none empty."""
    lib = _lib.load()
    BF16 = _lib.dtype_code(torch.bfloat16)

    def plan(N, H, T, D, nb):
        a, b = (ctypes.c_int64 * 1)(), (ctypes.c_int64 * 1)()
        assert lib.sea_performer_plan(N, H, T, D, nb, BF16, a, b) == 0
        return int(a[0]), int(b[0])

    assert plan(8, 32, 4096, 64, 33) == (1, 0)                      # headline batch: 256 pairs, one pass
    assert plan(4, 40, 4096, 128, 77) == (1, 0)                     # 160 pairs > 128: not cut
    nseg, ws = plan(1, 32, 8192, 80, 43)                            # BASELINE config 4, one sequence per GPU
    assert nseg == 8 and ws > 0 and ws % 16 == 0
    nseg5, ws5 = plan(1, 40, 4096, 128, 77)                         # config 5: 40 pairs -> 6 segments (240 workgroups)
    assert nseg5 == 6 and ws5 > 0
    assert plan(1, 4, 300, 64, 33) == (1, 0)                        # short sequence: fewer than 8 chunks
    for T in (513, 1000, 2049, 5000):
        n, _ = plan(1, 2, T, 64, 33)
        chunks = (T + 63) // 64
        seg_len = ((chunks + n - 1) // n) * 64
        assert 1 <= n <= 16 and (n - 1) * seg_len < T <= n * seg_len, (T, n)
    a, b = (ctypes.c_int64 * 1)(), (ctypes.c_int64 * 1)()
    assert lib.sea_performer_plan(1, 1, 1024, 96, 40, BF16, a, b) == -2      # SEA_EUNSUPPORTED head size
    assert b"D=96" in lib.sea_last_error()
    sb = lib.sea_performer_state_bytes(2, 4, 64, 33, BF16)
    assert sb > 0 and sb % (2 * 4) == 0 and lib.sea_performer_state_bytes(2, 4, 96, 33, BF16) == 0
    # algorithmic bytes of the graded kernel: Z (2 d s + 4) + N H T (2 d s + 4)   
    assert lib.sea_sparse_attention_bytes(1000, 2, 3, 10, 64, 2) == 1000 * (2 * 64 * 2 + 4) + 2 * 3 * 10 * (2 * 64 * 2 + 4)
