#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's own operators.

Run ONLY in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

It imports the reference's kernel modules
(`/root/reference/src/models/perlin_attention/ops/kernels/*.py`) IN PLACE and
runs them on the CPU through Triton's interpreter.  Nothing from the reference
is copied into this repository: the only things written are `.npz` files that
hold the seeded inputs and the outputs the reference produced for them.

Harness conditions (SURVEY.md "Facts established", §8c):
  * parent packages `src`, `src.models`, `src.models.perlin_attention`, `...ops`,
    `...ops.kernels` are pre-seeded as empty namespace stubs so that
    `perlin_attention/__init__.py` (which needs numba / performer_pytorch /
    transformers==4.32) never executes;
  * `TRITON_INTERPRET=1`;
  * Triton 3.6 has no `tl.math.round`; Triton 2.0 (the reference's pin) lowered
    it to libdevice `roundf` = round-half-away-from-zero, so the harness
    installs exactly that;
  * the reference writes masks as `a and b` between tensors
    (`causal_resize_m_to_t.py:541-571`, `flat_csr_elmul.py:82-108`).  Triton
    2.0's code generator lowered a Python BoolOp to `logical_and/logical_or`;
    Triton 3.6's *interpreter* executes the kernel body as plain Python, where
    `a and b` evaluates to `b` -- the masks degrade and masked stores run past
    their rows (observed: heap corruption).  The harness therefore extends the
    interpreter's own AST pass so that BoolOp means element-wise and/or again.

Cases follow the reference's own kernel self-tests
(`flat_csr_masked_bmm.py:207-323` etc.): probs = softmax(randn), causal mask,
q/k/v = randn, seed 42 -- at sizes the interpreter finishes in seconds.
"""
import os
import sys
import types
import math

os.environ["TRITON_INTERPRET"] = "1"

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch
import triton
import triton.language as tl


def _install_round_shim():
    if hasattr(tl.math, "round"):
        return
    @triton.jit
    def _round_half_away(x):
        return tl.where(x >= 0, tl.floor(x + 0.5), tl.ceil(x - 0.5))
    tl.math.round = _round_half_away


def _install_boolop_shim():
    import ast
    from triton.runtime import interpreter as _interp

    class _BoolOpAsLogical(_interp.ASTTransformer):
        def visit_Assign(self, node):
            # the stock pass does not descend into the assigned value
            node.value = self.visit(node.value)
            return super().visit_Assign(node)

        def visit_BoolOp(self, node):
            self.generic_visit(node)
            op = ast.BitAnd() if isinstance(node.op, ast.And) else ast.BitOr()
            expr = node.values[0]
            for v in node.values[1:]:
                expr = ast.BinOp(left=expr, op=op, right=v)
            return ast.copy_location(expr, node)

    _interp.FunctionRewriter.ast_transformer = _BoolOpAsLogical()


def _stub_packages():
    sys.path.insert(0, REF)
    chain = [
        ("src", "src"),
        ("src.models", "src/models"),
        ("src.models.perlin_attention", "src/models/perlin_attention"),
        ("src.models.perlin_attention.ops", "src/models/perlin_attention/ops"),
        ("src.models.perlin_attention.ops.kernels", "src/models/perlin_attention/ops/kernels"),
    ]
    for name, rel in chain:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, rel)]
        m.__package__ = name
        sys.modules[name] = m


def load_reference_ops():
    _install_round_shim()
    _install_boolop_shim()
    _stub_packages()
    import importlib
    K = "src.models.perlin_attention.ops.kernels."
    mods = {}
    for n in ["causal_topk_masking", "resize_m_to_t", "causal_resize_m_to_t",
              "flat_csr_masked_bmm", "flat_csr_softmax", "flat_csr_elmul",
              "flat_csr_sdbmm", "flat_csr_to_dense"]:
        mods[n] = importlib.import_module(K + n)
    return mods


FP_MIN = torch.finfo(torch.float16).min * 0.5

# (name, N, H, T_DST, T_SRC, T_M, k, d, causal)
CASES = [
    ("tiny",      1, 2, 32, 32, 8, 4, 16, True),
    ("mid",       2, 3, 96, 96, 16, 8, 16, True),
    ("ragged",    1, 3, 50, 50, 16, 8, 8, True),     # T % T_M != 0
    ("short",     1, 2, 12, 12, 16, 4, 8, True),     # T < T_M  (zero-width pixels)
    ("big",       1, 4, 128, 128, 32, 8, 32, True),
    ("clamp",     1, 2, 96, 96, 4, 8, 8, True),      # T/T_M = 24 > k = 8 -> max_k clamp fires
    ("noncausal", 1, 2, 40, 40, 8, 4, 8, False),     # F8, completeness only
    ("large",     1, 4, 512, 512, 64, 16, 32, True), # round 5: the largest case the interpreter finishes in minutes (T / T_M = 8)
]


def run_case(mods, name, N, H, T_DST, T_SRC, T_M, k, d, causal):
    torch.manual_seed(42)
    scores = torch.randn((N, H, T_DST, T_M))
    probs = torch.softmax(scores, dim=-1)
    T = T_SRC
    causal_attention_mask = ((torch.arange(T).view(1, T) > torch.arange(T).view(T, 1)) * FP_MIN).view(1, 1, T, T)
    # causal_topk_masking.py:26 views the causal mask as batch 1; the dense twin wants (N,1,T,T)
    causal_attention_mask_1 = causal_attention_mask[:, :, -T_DST:, :]
    causal_attention_mask = causal_attention_mask_1.expand(N, 1, T_DST, T).contiguous()
    if causal:
        attention_mask = causal_attention_mask[:, :, -1:, :]
        dst_attention_mask = causal_attention_mask[:, :, :, :1]
    else:
        attention_mask = torch.zeros((N, 1, 1, T))
        dst_attention_mask = torch.zeros((N, 1, T_DST, 1))

    out = dict(probs=probs.numpy(), meta=np.array([N, H, T_DST, T_SRC, T_M, k, d, int(causal)], dtype=np.int64))

    # F1: kernel-test top-k helper (causal_topk_masking.py:3-77)
    mask_m = mods["causal_topk_masking"].causal_topk_masking(
        probs, k=k, attention_mask=attention_mask, dst_attention_mask=dst_attention_mask,
        causal_attention_mask=causal_attention_mask_1, is_causal=causal)
    out["mask_m"] = mask_m.contiguous().numpy()

    # F2: dense twin (resize_m_to_t.py:6-73)
    if causal:
        dense = mods["resize_m_to_t"].resize_from_m_to_t(
            mask_m, 0, causal_attention_mask, target_width=T, training=False, is_causal=True, k=k, oversampled=1.0)
        # the module masks the causal part afterwards (attention.py:958-959)
        dense = dense.masked_fill(causal_attention_mask < -1, 0)
    else:
        dense = mods["resize_m_to_t"].resize_from_m_to_t(
            mask_m, 0, attention_mask, target_width=T, training=False, is_causal=False, k=k, oversampled=1.0)
    out["mask_dense"] = dense.contiguous().numpy()

    # F3: flat CSR (causal_resize_m_to_t.py:910-1007)
    csr = mods["causal_resize_m_to_t"].resize_from_m_to_t_csr(
        mask_m, 0, k, target_width=T, is_causal=causal)
    out["crow"] = csr.crow_indices().numpy()
    out["col"] = csr.col_indices().numpy()
    out["csr_dense"] = mods["flat_csr_to_dense"].flat_csr_to_dense(csr, T, H).numpy()

    q = torch.randn((N, H, T_DST, d))
    kk = torch.randn((N, H, T, d))
    v = torch.randn((N, H, T, d))
    scaler = torch.sigmoid(torch.randn((N, H, T_DST)))
    out.update(q=q.numpy(), k=kk.numpy(), v=v.numpy(), scaler=scaler.numpy())

    # F4..F7
    s = mods["flat_csr_masked_bmm"].flat_csr_masked_bmm(q, kk, csr)
    out["sddmm"] = s.values().numpy()
    p = mods["flat_csr_softmax"].flat_csr_softmax(s, H, T)
    out["softmax"] = p.values().numpy()
    e = mods["flat_csr_elmul"].flat_csr_elmul(p, scaler.view(N, H, T_DST, 1).expand(N, H, T_DST, T))
    out["elmul"] = e.values().numpy()
    o = mods["flat_csr_sdbmm"].flat_csr_sdbmm(e, v, T_M)
    out["sdbmm"] = o.numpy()

    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    nnz = int(out["crow"][:, -1].max())
    print(f"{name}: nnz={nnz} mask_m sum={out['mask_m'].sum():.0f} "
          f"dense==csr_dense: {np.array_equal((out['mask_dense'] > 0), (out['csr_dense'] > 0))} -> {path}")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; golden vectors can only be generated in the build container")
    mods = load_reference_ops()
    only = sys.argv[1:] or None
    for c in CASES:
        if only and c[0] not in only:
            continue
        run_case(mods, *c)


if __name__ == "__main__":
    main()
