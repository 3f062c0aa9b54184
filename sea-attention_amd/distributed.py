"""Multi-GPU: batch sharding of the SEA layer + one all-gather of outputs.

Every step of the hot path is independent per batch item, so the N sequences are split over the
ranks (one process per GPU, weights replicated) and the only exchange is ONE all-gather of the
`context_layer` shards (N/G, T, H*d) -- RCCL over xGMI on the GPU box (`backend="nccl"`), gloo in
the CPU tests.  Heads cannot be sharded for the predictor/top-k (the CNN mixes heads and the top-k
pools them, attention.py:271-276,844), so there is no tensor-parallel variant here.
"""
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `n_items` batch items; the first (n % world) ranks get one extra."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(t: torch.Tensor, world_size: int = None, rank: int = None) -> torch.Tensor:
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(t.shape[0], world_size, rank)
    return t[lo:hi]


def all_gather_context(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """Gather the per-rank (n_local, T, H*d) outputs into the full (N, T, H*d) tensor on every rank.
    Equal shards use one `all_gather_into_tensor` (a single fused RCCL all-gather); ragged shards fall
    back to a padded gather."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_bounds(n_items, world, r) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    local = local.contiguous()
    if len(set(counts)) == 1:
        out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(counts)
    padded = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    bufs: List[torch.Tensor] = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


class ContextGatherer:
    """Pipelined all-gather of the per-rank context shards: the collective of step i runs on the process group's
    own stream (RCCL over xGMI) while step i+1 computes.

    `depth` (default 2) slots of (local shard buffer, gathered buffer).  Protocol per step:
        slot = g.next_slot()            # the launch stream now waits for whatever last used this slot
        ... producer writes g.local[slot]  (e.g. the attention kernel's `out=`) ...
        full = g.launch(slot)           # asynchronous all-gather of g.local[slot] into g.out[slot]
    and `g.finish()` before the results are read / the timed region ends.  With world size 1 it degenerates to
    returning the local buffer.  Equal shards only (the ragged case uses `all_gather_context`)."""

    def __init__(self, local_shape, n_items: int, dtype, device, group=None, depth: int = 2):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        assert n_items % self.world == 0 and local_shape[0] * self.world == n_items, "equal shards only"
        self.depth = depth
        self.local = [torch.empty(tuple(local_shape), dtype=dtype, device=device) for _ in range(depth)]
        self.out = ([torch.empty((n_items,) + tuple(local_shape[1:]), dtype=dtype, device=device) for _ in range(depth)]
                    if self.world > 1 else self.local)
        self._work = [None] * depth
        self._i = 0
        self._sync_only = False

    def next_slot(self) -> int:
        slot = self._i % self.depth
        self._i += 1
        w = self._work[slot]
        if w is not None:                 # the collective that read local[slot] / wrote out[slot] `depth` steps ago
            w.wait()                      # nccl: a stream-level dependency, the host does not block
            self._work[slot] = None
        return slot

    def launch(self, slot: int) -> torch.Tensor:
        if self.world > 1:
            if not self._sync_only:
                try:
                    self._work[slot] = dist.all_gather_into_tensor(self.out[slot], self.local[slot], group=self.group,
                                                                    async_op=True)
                    return self.out[slot]
                except (RuntimeError, NotImplementedError):   # backend without the fused / asynchronous form: every
                    self._sync_only = True                    # rank takes the same branch (same software everywhere)
            chunks = list(self.out[slot].chunk(self.world, dim=0))
            dist.all_gather(chunks, self.local[slot], group=self.group)
        return self.out[slot]

    def finish(self) -> None:
        for s, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[s] = None


# ---- N < G: split the QUERY ROWS of steps G..L (secondary partitioning) -----------------------------------
def row_shard_bounds(T_dst: int, world_size: int, rank: int, k: int = 64, T_src: int = None) -> Tuple[int, int]:
    """Contiguous block of query rows for `rank`, balanced by the entries a causal row keeps: a row of width w
    (= absolute position + 1) emits about min(w, k) keys per head, so the early rows are cheaper and the first
    ranks get more of them.  Every step from the top-k on depends only on the row's own probabilities plus K/V
    up to that row, so the blocks are independent; K/V are replicated (2*H*T*d elements, small next to the work)."""
    T_src = T_dst if T_src is None else T_src
    w = torch.arange(T_src - T_dst + 1, T_src + 1, dtype=torch.float64)
    cost = torch.clamp_max(w, float(k)).cumsum(0)
    total = float(cost[-1])
    cuts = [0]
    for r in range(1, world_size):
        cuts.append(int(torch.searchsorted(cost, torch.tensor(total * r / world_size, dtype=torch.float64)).item()))
    cuts.append(T_dst)
    for i in range(1, len(cuts)):                      # monotone, never empty while rows remain
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def sparse_rows(ops, probs: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, lo: int, hi: int, top_k: int,
                row_scale: torch.Tensor = None, avg: torch.Tensor = None, mix: torch.Tensor = None, out_dtype=None):
    """Steps H..L for query rows [lo, hi) of a causal layer on the HIP kernels: the block is the LAST hi-lo rows of
    the hi-long prefix, which is exactly the kernels' `T_dst < T_src` contract (row width = T_src - T_dst + t + 1).
    probs (N,H,T,T_m) [or already sliced to the block], q (N,H,T,d), k/v (N,H,T,d).  Returns (N, hi-lo, H*d)."""
    N, H, T, d = q.shape
    T_m = probs.shape[-1]
    pr = probs if probs.shape[-2] == hi - lo else probs[:, :, lo:hi]
    keep = ops.keep_table_causal(H, hi, T_m, top_k)[lo:hi].contiguous().to(q.device)
    csr, _ = ops.topk_to_csr(pr.contiguous(), keep, top_k, target_width=hi, is_causal=True)
    sl = lambda t: None if t is None else t[:, :, lo:hi].contiguous()
    ctx = torch.empty((N, hi - lo, H * d), dtype=out_dtype or torch.float32, device=q.device)
    ops.sparse_attention(q[:, :, lo:hi], k[:, :, :hi], v[:, :, :hi], csr, row_scale=sl(row_scale), avg=sl(avg), mix=sl(mix),
                         out=ctx.view(N, hi - lo, H, d).permute(0, 2, 1, 3))
    return ctx


def all_gather_rows(local: torch.Tensor, bounds: List[Tuple[int, int]], group=None) -> torch.Tensor:
    """Gather ragged row blocks (N, t_r, C) from every rank into (N, T, C): one padded all-gather."""
    world = dist.get_world_size(group)
    sizes = [hi - lo for lo, hi in bounds]
    mx = max(sizes)
    N, _, C = local.shape
    padded = torch.zeros((N, mx, C), dtype=local.dtype, device=local.device)
    padded[:, :local.shape[1]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:, :s] for b, s in zip(bufs, sizes)], dim=1)
