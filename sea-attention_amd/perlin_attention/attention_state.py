"""KV-cache decoding state of the SEA layer.

Role of the reference's `attention_state.py` (`PerlinAttentionState` with its three sub-states, threaded through
`PerlinAttention.forward(..., last_state=...)` when `pconfig.use_cache`, attention.py:391-439,527-572,630-646,
1229-1235): a later call that brings only the NEW query rows `T_DST <= T_SRC` (keys / values cover the whole prefix)
must produce the rows a full forward over `T_SRC` tokens would have produced.  Three quantities of the estimator
and epilogue look back along the sequence and are therefore carried:

* `PerformerState`  the causal linear attention's running sums  S = sum_s phi(k_s)^T [pos_s | v_s]  (nb x 2d) and
                    ksum = sum_s phi(k_s)  (nb), per (n, h), kept in float64 like the reference's cumsums (:84-98);
* `CnnWindowState`  the last rows of the predictor CNN's input: its two dilated causal 3x3 convolutions reach back
                    2 * 2 * (3-1) = 8 rows, so 8 cached rows make the windowed CNN equal the full one (the
                    reference keeps a 24-row window, :146-147);
* `CumAvgState`     the running sum of v and the number of rows seen (:205-236).

The state is immutable from the caller's point of view: every forward returns a NEW `PerlinAttentionState`
(`clone()` shares tensors; sub-states replace, never mutate, their tensors), as the reference's copy-on-write
`get_cloned_state` does, so a caller may branch decoding from an earlier state.

Unlike the reference's `StatefulCausalPerformer.__call__` (which is handed q/k BEFORE the feature map and skips it,
a known quirk guarded by PERLIN_HOTFIX_STATEFUL), this restatement applies the same generalized-ReLU feature map as
the stateless path, so cached decoding reproduces the stateless rows (tests/test_kv_cache.py).
"""
from typing import Optional

import torch


class PerformerState:
    def __init__(self):
        self.S: Optional[torch.Tensor] = None      # (N, H, nb, e) float64
        self.ksum: Optional[torch.Tensor] = None   # (N, H, nb)    float64
        # HIP estimator (16-bit inference, d = 64): the same sums -- and the column sums of v -- as the opaque fp32
        # image `sea_performer_causal_step` reads and writes (the kernel's own accumulators); S / ksum stay None
        self.image: Optional[torch.Tensor] = None
        self.seq_index = 0

    def step(self, qp: torch.Tensor, kp: torch.Tensor, v: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
        """qp, kp (N,H,T_new,nb) feature maps of the NEW rows, v (N,H,T_new,e).  Returns ctx (N,H,T_new,e) float32."""
        assert self.image is None, "this state continues on the HIP Performer"
        qd, kd, vd = qp.double(), kp.double(), v.double()
        S0 = self.S if self.S is not None else torch.zeros(kd.shape[:2] + (kd.shape[-1], vd.shape[-1]), dtype=torch.float64, device=kd.device)
        k0 = self.ksum if self.ksum is not None else torch.zeros(kd.shape[:2] + (kd.shape[-1],), dtype=torch.float64, device=kd.device)
        a = torch.matmul(qd, kd.transpose(-1, -2)).tril_()                      # (N,H,T_new,T_new) in-chunk scores
        num = torch.matmul(a, vd) + torch.matmul(qd, S0)
        den = a.sum(-1) + (qd * k0.unsqueeze(-2)).sum(-1) + eps * qd.sum(-1)
        new = PerformerState()
        new.S = S0 + torch.matmul(kd.transpose(-1, -2), vd)
        new.ksum = k0 + kd.sum(-2)
        new.seq_index = self.seq_index + qp.shape[-2]
        return new, (num / den.unsqueeze(-1)).float()


def cnn_lookback(cnn) -> int:
    """Rows the predictor CNN reaches back along the sequence: sum of dilation * (kernel_size - 1) over the causal
    convolutions of its KeepRes body -- 2 convs * 2 * (3 - 1) = 8 for the standard predictor, 12 with
    PERLIN_HOTFIX_OPT_DEEPER=1 (three convs; attention.py:266-281).  The reference keeps a fixed 24-row window (:146-147)."""
    from .modules import CausalConv2d
    reach = 0
    for m in cnn.modules():
        if isinstance(m, CausalConv2d) and m.causal:
            dil = m.dilation if isinstance(m.dilation, int) else m.dilation[0]
            reach += dil * (m.kernel_size - 1)
    return max(reach, 1)


class CnnWindowState:
    LOOKBACK = 8          # the standard predictor's reach (see cnn_lookback); states are built with the real one

    def __init__(self, lookback: int = LOOKBACK):
        self.lookback = lookback
        self.rows: Optional[torch.Tensor] = None   # (N, C, <=lookback, W) trailing rows of the CNN input
        # HIP estimator (16-bit inference): the same window AFTER the CNN's first LayerNorm, channel-blocked
        # (N, <=lookback, C/8, W, 8), as the one-launch predictor MLP emits it; exactly one of the two is set
        self.rows_c8: Optional[torch.Tensor] = None

    def step(self, cnn, x: torch.Tensor):
        """x (N, C, T_new, W): the CNN input rows of the new tokens.  Returns (new_state, y (N, C', T_new, W'))."""
        T_new = x.shape[-2]
        xs = x if self.rows is None else torch.cat([self.rows, x], dim=-2)
        y = cnn(xs)[..., -T_new:, :]
        new = CnnWindowState(self.lookback)
        new.rows = xs[..., -self.lookback:, :].detach()
        return new, y


class CumAvgState:
    def __init__(self):
        self.cumsum: Optional[torch.Tensor] = None   # (N, H, 1, D) float32
        self.prev_len = 0
        self.in_image = False                        # HIP estimator: the sums live in PerformerState.image

    def step(self, v_new: torch.Tensor):
        """v_new (N,H,T_new,D).  Returns (new_state, cumulative average rows (N,H,T_new,D) in v's dtype)."""
        assert not self.in_image, "this state continues on the HIP Performer"
        cs = v_new.float().cumsum(-2)
        if self.cumsum is not None:
            cs = cs + self.cumsum
        T_new = v_new.shape[-2]
        den = torch.arange(self.prev_len + 1, self.prev_len + 1 + T_new, device=v_new.device, dtype=torch.float32).view(1, 1, -1, 1)
        new = CumAvgState()
        new.cumsum = cs[..., -1:, :].clone()
        new.prev_len = self.prev_len + T_new
        return new, (cs / den).to(v_new.dtype)


class PerlinAttentionState:
    """Per-layer decoding state; `states` maps the reference's names to the three sub-states."""
    PERFORMER = "performer->performer_context_layer"
    CNN = "attention_predictor_cnn->estimated_attention_score"
    CUMAVG = "output->cumavg"

    def __init__(self, parent=None):
        if parent is not None:
            self.num_heads = parent.num_attention_heads
            self.head_dim = parent.attention_head_size
            self.embd_dim = parent.all_head_size
        self.states = {}

    def clone(self) -> "PerlinAttentionState":
        new = PerlinAttentionState(None)
        for a in ("num_heads", "head_dim", "embd_dim"):
            if hasattr(self, a):
                setattr(new, a, getattr(self, a))
        new.states = dict(self.states)
        return new

    def get(self, name: str, factory):
        return self.states.get(name) or factory()

    @property
    def seq_len(self) -> int:
        c = self.states.get(self.CUMAVG)
        return c.prev_len if c is not None else 0

    def strify(self) -> str:
        return f"State(len={self.seq_len}, {sorted(self.states)})"
