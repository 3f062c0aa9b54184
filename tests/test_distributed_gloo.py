"""N>1 path on CPU: batch sharding + all-gather of context shards, world_size 2 over gloo.
The compute inside each rank is the CPU oracle (tests may use it as a stand-in checker)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_items, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sea_attention_amd import distributed as D
    from oracle import sea_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7)                                  # same full batch on every rank
    N, H, T, T_M, k, d = n_items, 2, 32, 8, 4, 8
    probs = torch.softmax(torch.randn(N, H, T, T_M), -1)
    q, kk, v = torch.randn(N, H, T, d), torch.randn(N, H, T, d), torch.randn(N, H, T, d)

    def layer(p_, q_, k_, v_):
        keep = O.keep_counts_module(H, T, T_M, k)
        crow, col = O.resize_m_to_t_csr(O.grouped_topk_mask(p_, keep), k, T, True)
        o = O.sparse_attention(q_, k_, v_, crow, col)
        return o.permute(0, 2, 1, 3).reshape(o.shape[0], T, H * d)

    full = layer(probs, q, kk, v)
    local = layer(*[D.shard_batch(t) for t in (probs, q, kk, v)])
    lo, hi = D.shard_bounds(N, world, rank)
    assert local.shape[0] == hi - lo
    gathered = D.all_gather_context(local, N)
    ok = torch.allclose(gathered, full, atol=1e-6)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


@pytest.mark.parametrize("n_items", [4, 5])               # equal shards and ragged shards
def test_shard_and_all_gather_world2(n_items):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_items, None), nprocs=2, join=True)


def test_shard_bounds_cover_everything():
    from sea_attention_amd.distributed import shard_bounds
    for n in (1, 7, 8, 9, 64):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker_pipelined(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sea_attention_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local, T, C, steps = 3, 5, 4, 7
    g = D.ContextGatherer((n_local, T, C), n_local * world, torch.float32, "cpu")
    ok = True
    pending = []
    for i in range(steps):
        slot = g.next_slot()                              # waits for the collective that used this slot 2 steps ago
        if len(pending) == g.depth:                       # ... whose result is now complete: check it
            j, full = pending.pop(0)
            exp = torch.cat([torch.full((n_local, T, C), float(100 * r + j)) for r in range(world)])
            ok &= torch.equal(full, exp)
        g.local[slot].fill_(float(100 * rank + i))        # "producer": this step's shard
        pending.append((i, g.launch(slot)))
    g.finish()
    for j, full in pending:
        exp = torch.cat([torch.full((n_local, T, C), float(100 * r + j)) for r in range(world)])
        ok &= torch.equal(full.clone(), exp)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def test_pipelined_context_gatherer_world2():
    """Double-buffered asynchronous all-gather (the multi-GPU bench path): every step's gathered tensor is right."""
    port = _free_port()
    mp.spawn(_worker_pipelined, args=(2, port, None), nprocs=2, join=True)


def test_context_gatherer_single_process_is_identity():
    from sea_attention_amd.distributed import ContextGatherer
    g = ContextGatherer((2, 3, 4), 2, torch.float32, "cpu")
    s = g.next_slot(); g.local[s].fill_(5.0)
    assert g.launch(s) is g.local[s]
    g.finish()


def test_row_shard_bounds_balance_the_kept_entries():
    from sea_attention_amd.distributed import row_shard_bounds
    T, k = 4096, 64
    for w in (2, 4, 8):
        spans = [row_shard_bounds(T, w, r, k) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == T and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        cost = torch.clamp_max(torch.arange(1, T + 1, dtype=torch.float64), k)
        loads = [float(cost[lo:hi].sum()) for lo, hi in spans]
        assert max(loads) / (sum(loads) / w) < 1.02
        assert spans[0][1] - spans[0][0] >= spans[-1][1] - spans[-1][0]      # early (cheap) rows: longer block


def _worker_rows(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sea_attention_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, C = 37, 6
    full = torch.arange(2 * T * C, dtype=torch.float32).view(2, T, C)
    bounds = [D.row_shard_bounds(T, world, r, k=8) for r in range(world)]
    lo, hi = bounds[rank]
    got = D.all_gather_rows(full[:, lo:hi].contiguous(), bounds)
    ok = torch.equal(got, full)
    dist.barrier(); dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def test_all_gather_of_ragged_row_blocks_world2():
    port = _free_port()
    mp.spawn(_worker_rows, args=(2, port, None), nprocs=2, join=True)


# ---- row-split whole layer: the protocol (increments -> prefix in rank order, window hand-off, ragged row gather) -------
def _toy_phases(x):
    """A stand-in with the layer's dependency structure: a causal prefix state (sum of all earlier rows) and an 8-row causal
    window (mean of the last 8 'CNN input' rows), rows (N, T, C)."""
    LB = 8

    def phase_a(lo, hi):
        return x[:, lo:hi].sum(1)                                       # additive state increment (N, C)

    def phase_b(lo, hi, state_in, hook):
        xr = x[:, lo:hi]
        prefix = torch.cumsum(xr, 1) + (state_in.unsqueeze(1) if state_in is not None else 0.0)
        feat = torch.tanh(prefix) * 0.5 + xr                            # "MLP output": the window-carrying rows
        halo = hook(feat)
        ext = feat if halo is None else torch.cat([halo, feat], 1)
        pad = torch.cat([torch.zeros_like(ext[:, :LB]), ext], 1)        # rows before the sequence read as zeros
        win = torch.stack([pad[:, i:i + ext.shape[1]] for i in range(LB + 1)], 0).sum(0)
        return (win[:, -xr.shape[1]:] * 0.1 + prefix).contiguous()
    return phase_a, phase_b, LB


def _worker_row_split(rank, world, port, T, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sea_attention_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    x = torch.randn(2, T, 5, dtype=torch.float64)
    pa, pb, lb = _toy_phases(x)
    full = pb(0, T, None, lambda f: None)                               # unsharded
    cuts = D.estimator_row_cuts(T, world, chunk=16)
    got = D.run_row_split(pa, pb, cuts, lb)
    local = D.run_row_split_local(pa, pb, cuts, lb)
    ok = torch.allclose(got, full, atol=1e-12) and torch.allclose(local, full, atol=1e-12)
    dist.barrier(); dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


@pytest.mark.parametrize("world,T", [(2, 96), (4, 200), (3, 112)])
def test_row_split_protocol_reproduces_the_unsharded_rows(world, T):
    """State increments all-gathered and prefix-summed in rank order, the window handed r -> r+1, ragged rows gathered: the
    sharded result IS the unsharded one (toy estimator with the layer's dependency structure; the real layer runs the same
    schedule on the GPU, tests/test_gpu_row_split.py)."""
    port = _free_port()
    mp.spawn(_worker_row_split, args=(world, port, T, None), nprocs=world, join=True)


def test_estimator_row_cuts_are_whole_chunks():
    from sea_attention_amd.distributed import estimator_row_cuts
    for T, w in ((4096, 8), (8192, 8), (4096, 3), (1000, 4), (64, 4)):
        cuts = estimator_row_cuts(T, w)
        assert cuts[0][0] == 0 and max(hi for _, hi in cuts) == T
        assert all(a[1] == b[0] or b[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert all(lo % 64 == 0 for lo, hi in cuts if hi > lo)


# ---- round 4: the all-gather cut into chunks of sequences -------------------------------------------------
def _chunked_worker(rank, world, port, chunks, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sea_attention_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    NB, T, C = 4, 6, 5
    g = D.ChunkedContextGatherer((NB, T, C), NB * world, torch.float32, "cpu", chunks=chunks)
    ref = D.ContextGatherer((NB, T, C), NB * world, torch.float32, "cpu")
    ok = True
    value = lambda r, step, i: float(1000 * r + 10 * step + i)          # item i of rank r at step `step`
    for step in range(5):                                               # more steps than slots: the slots are re-used
        slot = g.next_slot()
        for c in range(chunks):                                         # producer of chunk c, then its collective
            loc = g.local_chunk(slot, c)
            assert loc.is_contiguous() and tuple(loc.shape) == (NB // chunks, T, C)
            for i in range(g.n_c):
                loc[i].fill_(value(rank, step, c * g.n_c + i))
            g.launch_chunk(slot, c)
        rslot = ref.next_slot()
        for i in range(NB):
            ref.local[rslot][i].fill_(value(rank, step, i))
        full_ref = ref.launch(rslot)
        g.finish(); ref.finish()
        view = g.gathered(slot)                                         # (world, chunks, n_c, T, C)
        assert tuple(view.shape) == (world, chunks, NB // chunks, T, C)
        for r in range(world):
            for i in range(NB):
                ok &= bool((view[r, i // g.n_c, i % g.n_c] == value(r, step, i)).all())
        ok &= bool(torch.equal(g.gathered_items(slot), full_ref))       # same items, same order as the one-piece gatherer
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


@pytest.mark.parametrize("world,chunks", [(2, 1), (2, 2), (2, 4), (3, 2)])
def test_chunked_gatherer_equals_the_one_piece_gatherer(world, chunks):
    port = _free_port()
    mp.spawn(_chunked_worker, args=(world, port, chunks, None), nprocs=world, join=True)


def test_nccl_debug_parser():
    from sea_attention_amd.distributed import parse_nccl_debug
    log = """
host:123:456 [0] NCCL INFO RCCL version 2.22.3+hip7.0 HEAD:abcdef
host:123:456 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC/read
host:123:456 [0] NCCL INFO Channel 01/0 : 0[0] -> 7[7] via P2P/IPC
host:123:456 [0] NCCL INFO Connected all rings
host:123:456 [0] NCCL INFO 16 coll channels, 16 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer
host:123:456 [0] NCCL INFO AllGather: 134217728 Bytes -> Algo 1 proto 2 time 1234.5
host:123:456 [0] NCCL INFO AllGather: opCount 5 sendbuff 0x1 recvbuff 0x2 count 67108864 datatype 9 op 0 root 0 comm 0x3 [nranks=8] stream 0x4 algorithm Ring protocol Simple
"""
    r = parse_nccl_debug(log)
    assert r["version"].startswith("2.22.3") and r["channels"] == 16
    assert "P2P/IPC/read" in r["transports"] and "P2P/IPC" in r["transports"]
    assert {"algorithm": "Ring", "protocol": "Simple"} in r["allgather"] and len(r["allgather"]) == 1
    assert parse_nccl_debug("nothing useful here") == {"version": None, "channels": None, "transports": [], "allgather": []}
