"""-m gpu: the HIP path (through the C ABI) against the golden vectors the reference produced.
Indices bit-exact (order-sensitive); floating point within 1e-5 abs of the reference's fp32 results."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES

pytestmark = pytest.mark.gpu


def _meta(g):
    N, H, T_DST, T_SRC, T_M, k, d, causal = [int(x) for x in g["meta"]]
    return N, H, T_DST, T_SRC, T_M, k, d, bool(causal)


def _keep(ops, g, dev):
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    if causal:
        return ops.keep_table_kernel_test(H, T_DST, T_M, k, device=dev)
    per = torch.clamp_min(H * torch.round(k * T_M / torch.full((N, 1), T_SRC)), 1).to(torch.int32)
    return per.expand(N, T_DST).contiguous().to(dev)


@pytest.fixture(scope="module")
def ops():
    from sea_attention_amd.perlin_attention import ops
    from sea_attention_amd import _lib
    assert _lib.load().sea_version() == _lib.ABI_VERSION
    return ops


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_topk_mask(golden, ops, case):
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    dev = torch.device("cuda:0")
    probs = torch.from_numpy(g["probs"]).to(dev)
    mask = ops.topk_mask(probs, _keep(ops, g, dev), k, target_width=T_SRC, is_causal=causal)
    assert np.array_equal(mask.cpu().numpy(), g["mask_m"])


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_fused_topk_to_csr_bit_exact(golden, ops, case):
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    dev = torch.device("cuda:0")
    probs = torch.from_numpy(g["probs"]).to(dev)
    csr, mask = ops.topk_to_csr(probs, _keep(ops, g, dev), k, target_width=T_SRC, is_causal=causal, want_mask=True)
    assert np.array_equal(mask.cpu().numpy(), g["mask_m"])
    assert np.array_equal(csr.crow.cpu().numpy().astype(np.int64), g["crow"])
    t = csr.to_sparse_csr()
    assert np.array_equal(t.crow_indices().cpu().numpy(), g["crow"])
    assert np.array_equal(t.col_indices().cpu().numpy(), g["col"])          # order + zero padding
    # per-(row, head) offsets against a recount from the golden columns
    from oracle import sea_oracle as O
    ho = O.head_offsets(torch.from_numpy(g["crow"]), torch.from_numpy(g["col"]), H, T_SRC)
    assert np.array_equal(csr.head_off.cpu().numpy().astype(np.int64), ho.numpy())


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_dropin_resize_csr(golden, ops, case):
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    dev = torch.device("cuda:0")
    csr = ops.resize_from_m_to_t_csr(torch.from_numpy(g["mask_m"]).to(dev), 0, k, target_width=T_SRC, is_causal=causal)
    assert csr.is_sparse_csr and tuple(csr.shape) == (N, T_DST, H * T_SRC)
    assert csr.crow_indices().dtype == torch.int64
    assert np.array_equal(csr.crow_indices().cpu().numpy(), g["crow"])
    assert np.array_equal(csr.col_indices().cpu().numpy(), g["col"])
    assert torch.all(csr.values() == 1)
    dense = ops.flat_csr_to_dense(csr, T_SRC, H).cpu()
    from oracle import sea_oracle as O
    ref = O.flat_csr_to_dense(torch.from_numpy(g["crow"]), torch.from_numpy(g["col"]), torch.ones(g["col"].shape), T_SRC, H)
    assert torch.equal(dense, ref)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_dropin_csr_operators(golden, ops, case):
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    dev = torch.device("cuda:0")
    crow, col = torch.from_numpy(g["crow"]).to(dev), torch.from_numpy(g["col"]).to(dev)
    valid = np.arange(g["col"].shape[1])[None, :] < g["crow"][:, -1:]
    mask = torch.sparse_csr_tensor(crow, col, torch.ones(col.shape, device=dev), size=(N, T_DST, H * T_SRC))
    q, kk, v = (torch.from_numpy(g[n]).to(dev) for n in ("q", "k", "v"))
    s = ops.flat_csr_masked_bmm(q, kk, mask)
    np.testing.assert_allclose(s.values().cpu().numpy(), g["sddmm"], atol=1e-5, rtol=1e-5)   # padding keeps 1.0
    s_ref = torch.sparse_csr_tensor(crow, col, torch.from_numpy(g["sddmm"]).to(dev), size=mask.shape)
    p = ops.flat_csr_softmax(s_ref, H, T_SRC)
    np.testing.assert_allclose(p.values().cpu().numpy()[valid], g["softmax"][valid], atol=1e-6, rtol=1e-5)
    p_ref = torch.sparse_csr_tensor(crow, col, torch.from_numpy(g["softmax"]).to(dev), size=mask.shape)
    scaler = torch.from_numpy(g["scaler"]).to(dev)
    e = ops.flat_csr_elmul(p_ref, scaler.view(N, H, T_DST, 1).expand(N, H, T_DST, T_SRC))
    np.testing.assert_allclose(e.values().cpu().numpy()[valid], g["elmul"][valid], atol=1e-7, rtol=1e-6)
    e_ref = torch.sparse_csr_tensor(crow, col, torch.from_numpy(g["elmul"]).to(dev), size=mask.shape)
    o = ops.flat_csr_sdbmm(e_ref, v, T_M)
    assert o.dtype == torch.float32 and tuple(o.shape) == (N, H, T_DST, d)
    np.testing.assert_allclose(o.cpu().numpy(), g["sdbmm"], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_fused_sparse_attention(golden, ops, case):
    """one fused launch == the reference's four-operator chain (golden `sdbmm`)."""
    g = golden(case)
    N, H, T_DST, T_SRC, T_M, k, d, causal = _meta(g)
    dev = torch.device("cuda:0")
    probs = torch.from_numpy(g["probs"]).to(dev)
    csr, _ = ops.topk_to_csr(probs, _keep(ops, g, dev), k, target_width=T_SRC, is_causal=causal)
    q, kk, v = (torch.from_numpy(g[n]).to(dev) for n in ("q", "k", "v"))
    scaler = torch.from_numpy(g["scaler"]).to(dev).contiguous()
    out = ops.sparse_attention(q, kk, v, csr, row_scale=scaler)
    np.testing.assert_allclose(out.cpu().numpy(), g["sdbmm"], atol=2e-5, rtol=1e-4)
    # fused mix epilogue + direct (N,T,H*D) layout
    from oracle import sea_oracle as O
    s1 = torch.randn(N, H, T_DST, generator=torch.Generator().manual_seed(1))
    avg = O.cumavg(torch.from_numpy(g["v"]))[:, :, -T_DST:].contiguous()
    ref = torch.from_numpy(g["sdbmm"]) * torch.sigmoid(s1).unsqueeze(-1) + (1 - torch.sigmoid(s1).unsqueeze(-1)) * avg
    ctx = torch.empty((N, T_DST, H * d), device=dev)
    ops.sparse_attention(q, kk, v, csr, row_scale=scaler, avg=avg.to(dev), mix=torch.sigmoid(s1).to(dev).contiguous(),
                         out=ctx.view(N, T_DST, H, d).permute(0, 2, 1, 3))
    np.testing.assert_allclose(ctx.cpu().numpy(), ref.permute(0, 2, 1, 3).reshape(N, T_DST, H * d).numpy(),
                               atol=2e-5, rtol=1e-4)
