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
            self._work[slot] = dist.all_gather_into_tensor(self.out[slot], self.local[slot], group=self.group, async_op=True)
        return self.out[slot]

    def finish(self) -> None:
        for s, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[s] = None
