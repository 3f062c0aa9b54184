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
        if self.world > 1:
            # in-place all-gather: the producer (the attention kernel's `out=`) writes this rank's shard AT ITS OFFSET of the
            # gathered buffer, and the collective is told so (input = the rank-th chunk of the output): no staging copy
            # of the 134 MB shard, neither here nor inside RCCL
            rank = dist.get_rank(group)
            self.out = [torch.empty((n_items,) + tuple(local_shape[1:]), dtype=dtype, device=device) for _ in range(depth)]
            self.local = [o[rank * local_shape[0]:(rank + 1) * local_shape[0]] for o in self.out]
        else:
            self.local = [torch.empty(tuple(local_shape), dtype=dtype, device=device) for _ in range(depth)]
            self.out = self.local
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


class ChunkedContextGatherer:
    """The same pipelined all-gather with the shard cut into `chunks` groups of sequences, so that the collective of a step
    can START while the step still computes: the attention launch of chunk c writes `local_chunk(slot, c)`, an event later
    `launch_chunk(slot, c)` sends it -- xGMI carries chunk 0 while the kernels of chunk 1 run (round 4, VERDICT r3 item 8:
    at G = 8 the 134 MB shard is ~1 GB received per GPU and step, as long as the compute step itself; cutting it does not
    shrink the bytes, it removes the serial tail 'last kernel -> first byte on the wire' of every step and lets RCCL's
    channels work on 1 / chunks of the data at a time).

    Buffer layout per slot: (chunks, world, n_c, *rest) -- every chunk's gathered result is ONE contiguous block, so each
    collective is an in-place `all_gather_into_tensor` (input = the rank-th part of its output, no staging copy).
    `gathered(slot)` is the (world, chunks, n_c, *rest) VIEW of it: item i of rank r lives at [r, i // n_c, i % n_c]; a
    consumer that works row-wise (the out-projection GEMM over the last dimension) takes the view as it is,
    `gathered_items(slot)` makes the (world * n_local, *rest) copy for anybody else."""

    def __init__(self, local_shape, n_items: int, dtype, device, chunks: int, group=None, depth: int = 2):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        n_local = local_shape[0]
        assert n_local * self.world == n_items, "equal shards only"
        assert chunks >= 1 and n_local % chunks == 0, "the shard is cut into equal groups of whole sequences"
        self.chunks, self.n_c, self.depth = chunks, n_local // chunks, depth
        rest = tuple(local_shape[1:])
        self.buf = [torch.empty((chunks, self.world, self.n_c) + rest, dtype=dtype, device=device) for _ in range(depth)]
        self._work = [[None] * chunks for _ in range(depth)]
        self._i = 0
        # capability decided ONCE, here, by a probe collective every rank issues alike: a backend without the fused in-place
        # form raises on all ranks together and all of them take the list form from then on.  Nothing is caught later -- a real
        # communication failure on one rank must propagate, not silently switch that rank to a different collective
        # (ranks issuing different collectives hang the job)
        self._sync_only = False
        if self.world > 1:
            probe_in = torch.zeros((1,), dtype=dtype, device=device)
            probe_out = torch.zeros((self.world,), dtype=dtype, device=device)
            try:
                dist.all_gather_into_tensor(probe_out, probe_in, group=group, async_op=True).wait()
            except NotImplementedError:
                self._sync_only = True
            except RuntimeError as e:
                if not any(w in str(e).lower() for w in ("not supported", "not implemented", "unsupported", "does not support")):
                    raise
                self._sync_only = True
            if self._sync_only and self.rank == 0:
                import warnings
                warnings.warn(f"ChunkedContextGatherer: backend {dist.get_backend(group)!r} has no asynchronous in-place all-gather; "
                              "every chunk takes the blocking list form (no overlap with the next chunk's attention launch)")

    def next_slot(self) -> int:
        slot = self._i % self.depth
        self._i += 1
        self._wait(slot)
        return slot

    def _wait(self, slot):
        for c, w in enumerate(self._work[slot]):
            if w is not None:
                w.wait()                      # nccl: a stream-level dependency, the host does not block
                self._work[slot][c] = None

    def local_chunk(self, slot: int, c: int) -> torch.Tensor:
        """(n_c, *rest), contiguous: where the producer writes sequences c * n_c .. (c + 1) * n_c - 1 of this rank's shard."""
        return self.buf[slot][c, self.rank]

    def launch_chunk(self, slot: int, c: int) -> None:
        """Asynchronous in-place all-gather of chunk c (call it once the producer of `local_chunk(slot, c)` is enqueued)."""
        if self.world == 1:
            return
        out = self.buf[slot][c].view((self.world * self.n_c,) + tuple(self.buf[slot].shape[3:]))
        if not self._sync_only:
            self._work[slot][c] = dist.all_gather_into_tensor(out, self.local_chunk(slot, c), group=self.group, async_op=True)
            return
        dist.all_gather(list(out.chunk(self.world, dim=0)), self.local_chunk(slot, c).clone(), group=self.group)

    def gathered(self, slot: int) -> torch.Tensor:
        """(world, chunks, n_c, *rest) view of the slot: [r, c, i] = sequence c * n_c + i of rank r."""
        return self.buf[slot].transpose(0, 1)

    def gathered_items(self, slot: int) -> torch.Tensor:
        """(world * n_local, *rest) copy in rank-major item order (what `ContextGatherer.out[slot]` holds)."""
        g = self.gathered(slot)
        return g.reshape((self.world * self.chunks * self.n_c,) + tuple(g.shape[3:]))

    def finish(self) -> None:
        for s in range(self.depth):
            self._wait(s)


def parse_nccl_debug(text: str) -> dict:
    """What RCCL / NCCL said it chose, from its NCCL_DEBUG=INFO (+ NCCL_DEBUG_SUBSYS=INIT,COLL,TUNING,GRAPH) log: version,
    transports of the rings' links, channel count, and the (algorithm, protocol) pairs its tuner picked for AllGather.
    Tolerant by design -- the log format differs between versions -- and never raises: unknown text gives empty fields."""
    import re
    algos = {"0": "Tree", "1": "Ring", "2": "CollNetDirect", "3": "CollNetChain", "4": "NVLS", "5": "NVLSTree"}
    protos = {"0": "LL", "1": "LL128", "2": "Simple"}
    out = {"version": None, "channels": None, "transports": [], "allgather": []}
    m = re.search(r"(?:NCCL|RCCL) version ([\w.+\-]+)", text)
    if m:
        out["version"] = m.group(1)
    m = re.search(r"(\d+) coll channels", text)
    if m:
        out["channels"] = int(m.group(1))
    tr = set(re.findall(r"via (P2P/[\w/]+|SHM[\w/]*|NET/[\w/]+|direct[\w/ ]*)", text))
    out["transports"] = sorted(t.strip() for t in tr)
    seen = set()
    for m in re.finditer(r"AllGather[^\n]*?[Aa]lgo(?:rithm)?\s*[:=]?\s*(\w+)[^\n]*?[Pp]roto(?:col)?\s*[:=]?\s*(\w+)", text):
        a, p = m.group(1), m.group(2)
        pair = (algos.get(a, a), protos.get(p, p))
        if pair not in seen:
            seen.add(pair)
            out["allgather"].append({"algorithm": pair[0], "protocol": pair[1]})
    return out


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


# ---- N < G, whole layer: split the query rows of EVERY step, estimator included ------------------------------
# A single long sequence (BASELINE config 4: one 8192-token sequence per GPU group) leaves batch sharding nothing to cut.
# `sparse_rows` above splits steps H..L and replicates the estimator (A..G = 59 % of the step at the headline shape);
# here the estimator is split too.  Rank r owns the rows [lo_r, hi_r) (equal cuts of whole 64-row Performer chunks) and
# needs from the ranks before it exactly what kv-cache decoding carries from the tokens before it:
#   * the causal Performer's running sums up to row lo_r.  They are linear in the rows, so every rank first runs the
#     Performer over its own rows FROM ZERO (phase A: only the state image matters), the images are all-gathered
#     (N*H x ~30 KB each) and rank r starts from inc_0 + ... + inc_{r-1}, added in rank order -- the same sum, in the same
#     order, the one-GPU sequence-parallel Performer forms (sea_performer_causal_segmented);
#   * the last 8 rows of the predictor CNN's input (two dilated causal 3-tap convolutions reach back 2*2*(3-1) rows):
#     one point-to-point hand-off r -> r+1 of the channel-blocked rows the one-launch MLP has just produced (64 KB);
#   * K and V of rows < hi_r: replicated (2*H*T*d elements).
# Phase B is then `PerlinAttention._forward_cached` on the rank's rows -- the decode path, T_dst < T_src -- with a state built
# from those two pieces, and one (ragged) all-gather of the context rows.  Nothing of the step is replicated except
# phase A's state-only Performer pass over the rank's own rows (about a third of a Performer launch).
def estimator_row_cuts(T: int, world: int, chunk: int = 64):
    """Equal cuts of whole Performer chunks: the cuts `sea_performer_plan` would make for `world` segments."""
    chunks = (T + chunk - 1) // chunk
    seg = ((chunks + world - 1) // world) * chunk
    return [(min(r * seg, T), min((r + 1) * seg, T)) for r in range(world)]


def _prefix_in_rank_order(incs, rank):
    """inc_0 + inc_1 + ... + inc_{rank-1}, added left to right (None for rank 0)."""
    acc = None
    for r2 in range(rank):
        if incs[r2] is not None:
            acc = incs[r2] if acc is None else acc + incs[r2]
    return acc


def run_row_split_local(phase_a, phase_b, cuts, lookback: int):
    """The row-split schedule with all ranks run one after another in THIS process.
    phase_a(lo, hi) -> additive state increment of the rows; phase_b(lo, hi, state_in, hook) -> (N, hi-lo, C) rows, calling
    hook(x) once with its fresh window-carrying rows x (rows on dim 1) to obtain the previous rank's last `lookback` rows."""
    incs = [phase_a(lo, hi) if hi > lo else None for lo, hi in cuts]
    tails, rows = {}, []
    for r, (lo, hi) in enumerate(cuts):
        if hi <= lo:
            continue

        def hook(x, r=r):
            tails[r] = x[:, -lookback:]
            return tails.get(r - 1)
        rows.append(phase_b(lo, hi, _prefix_in_rank_order(incs, r), hook))
    return torch.cat(rows, dim=1)


def run_row_split(phase_a, phase_b, cuts, lookback: int, group=None):
    """The same schedule across the ranks of `group`: all-gather of the state increments, prefix sum in rank order, one
    point-to-point hand-off r -> r+1 of the window rows, (ragged) all-gather of the result rows.  Every rank must own at
    least `lookback` rows (or none)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if any(hi_ - lo_ < lookback for lo_, hi_ in cuts):      # the same cuts on every rank: all of them raise together
        raise ValueError(f"row split over {world} ranks needs at least {lookback} rows per rank, cuts are {cuts}")
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    lo, hi = cuts[rank]
    inc = phase_a(lo, hi)
    incs = [torch.empty_like(inc) for _ in range(world)]
    dist.all_gather(incs, inc, group=group)

    def hook(x):
        tail = x[:, -lookback:].contiguous()
        req = None
        if rank + 1 < world:
            req = dist.isend(tail, peer(rank + 1), group=group)
        halo = None
        if rank > 0:
            halo = torch.empty_like(tail)
            dist.recv(halo, peer(rank - 1), group=group)
        if req is not None:
            req.wait()
        return halo
    local = phase_b(lo, hi, _prefix_in_rank_order(incs, rank), hook)
    return all_gather_rows(local.contiguous(), cuts, group=group)


def _phase_a_increment(attention, q, k, v, lo, hi):
    """State image of the Performer over rows [lo, hi) started from zero (phase A)."""
    from .perlin_attention import ops
    pos = attention.v_eye_learned_causal[0, 0, lo:, :]
    # (rows lo .. hi from zero: a step at t_base = 0 over the slices; cuts are whole Performer chunks, so the image it returns
    # -- the state at the last chunk boundary -- covers all of the rank's rows for every rank whose increment is used)
    _pv, _avg, image = ops.performer_step(q[:, :, lo:hi], k[:, :, lo:hi], v[:, :, lo:hi], pos,
                                          attention.performer.projection_matrix, state_in=None, t_base=0,
                                          n_segments=attention.performer_segments or 1)
    return image


def _phase_b_rows(layer, q, k, v, mask_rows, lo, hi, image_in, window):
    """The rank's rows through the whole layer (decode path) from the handed-over state; `window` is a hook called with the
    rank's fresh CNN input rows (channel-blocked, rows on dim 1) that returns the previous rank's last rows, or None."""
    from .perlin_attention.attention_state import (PerlinAttentionState, PerformerState, CnnWindowState, CumAvgState,
                                                   cnn_lookback)
    att = layer.attention
    st = PerlinAttentionState(att)
    if lo > 0:
        ps = PerformerState(); ps.image, ps.seq_index = image_in, lo
        cav = CumAvgState(); cav.prev_len, cav.in_image = lo, True
        st.states[PerlinAttentionState.PERFORMER] = ps
        st.states[PerlinAttentionState.CUMAVG] = cav
    cs = CnnWindowState(cnn_lookback(att.attention_predictor_cnn))
    cs.rows_c8 = window
    st.states[PerlinAttentionState.CNN] = cs
    out = layer(None, None, None, query_layer=q[:, :, lo:hi], key_layer=k[:, :, :hi], value_layer=v[:, :, :hi],
                attention_mask=mask_rows, last_state=st)
    return out.context_layer


def _layer_phases(layer, q, k, v, causal_mask_fn):
    from .perlin_attention.attention_state import cnn_lookback
    pa = lambda lo, hi: _phase_a_increment(layer.attention, q, k, v, lo, hi)
    pb = lambda lo, hi, st, hook: _phase_b_rows(layer, q, k, v, causal_mask_fn(lo, hi), lo, hi, st, hook)
    return pa, pb, cnn_lookback(layer.attention.attention_predictor_cnn)


def row_split_layer_local(layer, q, k, v, causal_mask_fn, world: int):
    """The row-split SEA layer of `world` ranks run one after another in this process (tests; a single GPU rehearsing the
    multi-GPU schedule).  causal_mask_fn(lo, hi) -> (N,1,hi-lo,hi) additive mask rows.  Returns the (N, T, H*d) context --
    the unsharded layer's."""
    pa, pb, lb = _layer_phases(layer, q, k, v, causal_mask_fn)
    return run_row_split_local(pa, pb, estimator_row_cuts(q.shape[-2], world), lb)


def row_split_layer(layer, q, k, v, causal_mask_fn, group=None):
    """The row-split SEA layer across the ranks of `group` (one process per GPU, RCCL over xGMI): every rank returns the full
    (N, T, H*d) context.  16-bit inference with d = 64 (the HIP estimator carries the state)."""
    pa, pb, lb = _layer_phases(layer, q, k, v, causal_mask_fn)
    return run_row_split(pa, pb, estimator_row_cuts(q.shape[-2], dist.get_world_size(group)), lb, group=group)
