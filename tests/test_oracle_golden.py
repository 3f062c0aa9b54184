"""Pin the CPU oracle (oracle/sea_oracle.py) to the reference's own outputs.

The golden vectors in tests/golden/*.npz were produced by running the reference's
operators in place (tests/golden/make_golden.py).  Integer/index results must be
bit-exact; floating-point results within 1e-5 abs (both sides accumulate in fp32,
only the summation order differs).
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES
from oracle import sea_oracle as O


def _meta(g):
    N, H, T_DST, T_SRC, T_M, k, d, causal = [int(x) for x in g["meta"]]
    return N, H, T_DST, T_SRC, T_M, k, d, bool(causal)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_topk_mask_kernel_variant(golden, case):
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    probs = torch.from_numpy(g["probs"])
    if causal:
        keep = O.keep_counts_kernel_test(H, T_DST, T_M, k)
    else:
        keep = O.keep_counts_kernel_test_noncausal(H, T_M, k, torch.full((N, 1), T_SRC)).view(N, 1)
    mask = O.grouped_topk_mask(probs, keep)
    assert np.array_equal(mask.numpy(), g["mask_m"])


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_dense_twin(golden, case):
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    mask_m = torch.from_numpy(g["mask_m"])
    fp_min = torch.finfo(torch.float16).min * 0.5
    if causal:
        cm = ((torch.arange(T_SRC).view(1, -1) > torch.arange(T_SRC).view(-1, 1)) * fp_min).view(1, 1, T_SRC, T_SRC)
        cm = cm[:, :, -T_DST:, :].expand(N, 1, T_DST, T_SRC)
        dense = O.resize_m_to_t_dense(mask_m, 0, cm, T_SRC, True).masked_fill(cm < -1, 0)
    else:
        dense = O.resize_m_to_t_dense(mask_m, 0, torch.zeros((N, 1, 1, T_SRC)), T_SRC, False)
    assert np.array_equal(dense.numpy(), g["mask_dense"])


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_csr_layout_bit_exact(golden, case):
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    crow, col = O.resize_m_to_t_csr(torch.from_numpy(g["mask_m"]), k, T_SRC, causal)
    assert np.array_equal(crow.numpy(), g["crow"])
    assert col.shape == g["col"].shape
    assert np.array_equal(col.numpy(), g["col"])          # order-sensitive, padding included


@pytest.mark.parametrize("case", ["tiny", "ragged", "short", "big"])
def test_csr_equals_dense_twin_when_unclamped(golden, case):
    """SURVEY 'facts': densified CSR == dense resize when the max_k clamp is idle."""
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    crow, col = torch.from_numpy(g["crow"]), torch.from_numpy(g["col"])
    dense = O.flat_csr_to_dense(crow, col, torch.ones(col.shape), T_SRC, H)
    assert dense.max() == 1                               # no duplicate columns (flat_csr_sdbmm.py:498-502)
    assert np.array_equal(dense.numpy() > 0, g["mask_dense"] > 0)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_csr_operators(golden, case):
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    crow, col = torch.from_numpy(g["crow"]), torch.from_numpy(g["col"])
    q, kk, v = (torch.from_numpy(g[n]) for n in ("q", "k", "v"))
    scaler = torch.from_numpy(g["scaler"])
    s = O.csr_sddmm(q, kk, crow, col)
    np.testing.assert_allclose(s.numpy(), g["sddmm"], atol=1e-5, rtol=1e-5)
    p = O.csr_softmax(torch.from_numpy(g["sddmm"]), crow, col, H, T_SRC)
    valid = np.arange(col.shape[1])[None, :] < g["crow"][:, -1:]
    np.testing.assert_allclose(p.numpy()[valid], g["softmax"][valid], atol=1e-6, rtol=1e-5)
    other = scaler.view(N, H, T_DST, 1).expand(N, H, T_DST, T_SRC)
    e = O.csr_elmul(torch.from_numpy(g["softmax"]), crow, col, other, T_SRC)
    np.testing.assert_allclose(e.numpy()[valid], g["elmul"][valid], atol=1e-7, rtol=1e-6)
    o = O.csr_spmm(torch.from_numpy(g["elmul"]), crow, col, v, T_SRC)
    np.testing.assert_allclose(o.numpy(), g["sdbmm"], atol=1e-5, rtol=1e-5)
    # and the composition
    o2 = O.sparse_attention(q, kk, v, crow, col, scaler)
    np.testing.assert_allclose(o2.numpy(), g["sdbmm"], atol=2e-5, rtol=1e-4)


def test_head_offsets_consistent(golden):
    g = golden("big")
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    crow, col = torch.from_numpy(g["crow"]), torch.from_numpy(g["col"])
    ho = O.head_offsets(crow, col, H, T_SRC)
    assert torch.equal(ho[:, :, -1], crow[:, 1:] - crow[:, :-1])
    # entries are grouped by ascending head inside every row (flat_csr_sdbmm.py:227-263 relies on it)
    for t in range(T_DST):
        seg = col[0, crow[0, t]:crow[0, t + 1]] // T_SRC
        assert torch.all(seg[1:] >= seg[:-1])


def test_module_keep_counts_match_fp32_formula():
    """K_t of the module path at every BASELINE config: no x.5 ties, fp32 quotient."""
    for H, T, T_M, k in [(12, 2048, 256, 64), (32, 4096, 256, 64), (32, 8192, 256, 64), (40, 4096, 256, 64)]:
        keep = O.keep_counts_module(H, T, T_M, k)
        assert keep.dtype == torch.float32
        t1 = np.arange(1, T + 1, dtype=np.float64)
        exact = H * k * T_M / t1
        assert np.all(np.abs(exact - np.floor(exact) - 0.5) > 1e-6)      # no rounding ties exist
        ref = np.maximum(np.rint(exact), 1)
        # fp32 quotient may differ from the exact one only where exact is within fp32 eps of x.5
        assert np.array_equal(keep.numpy().astype(np.int64), ref.astype(np.int64))


def test_dense_path_equals_sparse_composition(golden):
    """CPU-baseline restatement (dense branch) == sparse composition on the same mask."""
    g = golden("big")
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    probs = torch.from_numpy(g["probs"])
    q, kk, v = (torch.from_numpy(g[n]) for n in ("q", "k", "v"))
    scaler = torch.from_numpy(g["scaler"])
    out_dense, mask_m = O.dense_path(probs, q, kk, v, scaler, k)
    crow, col = O.resize_m_to_t_csr(mask_m, k, T_SRC, True)
    out_sparse = O.sparse_attention(q, kk, v, crow, col, scaler)
    np.testing.assert_allclose(out_dense.numpy(), out_sparse.numpy(), atol=2e-5, rtol=1e-4)
