#!/usr/bin/env python3
"""bench.py -- SEA sparse-attention layer on MI355X: tokens/s + achieved HBM GB/s of the sparse kernel.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run with one rank per GPU (RCCL).  One JSON line on rank 0.

A "step" = one forward of ONE SEA attention layer in sparse mode (`benchmarking=True`, steps A..L of
SURVEY.md section 3B: value augmentation, Performer, predictor MLP+CNN, softmax, HIP grouped top-k +
interpolation -> flat CSR, HIP fused sparse attention + mix) on a synthetic batch, followed -- when N>1 --
by the RCCL all-gather of the context shards.  Workload = BASELINE.json configs[2]:
OPT-1.3B SEA, H=32 d=64 T=4096, k=64, predictor length 256, nb-factor 8, bf16, random-init weights (seed 42).
Per-GPU batch is fixed (weak scaling).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "opt-1.3b": dict(H=32, d=64, T=4096, T_M=256, k=64, nbf=8),
    "opt-125m": dict(H=12, d=64, T=2048, T_M=256, k=64, nbf=8),
    "opt-2.7b": dict(H=32, d=80, T=8192, T_M=256, k=64, nbf=8),
    "llama-13b": dict(H=40, d=128, T=4096, T_M=256, k=64, nbf=8),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_GATHER_GBS = 17800.0    # MI355X_MICROARCH.md "Indexed rows": 16.8-18.8 TB/s chip-wide for rows served by the XCD's L2


def _attn_source_sha():
    """sha256 over the sources of the graded kernel: a PMC traffic record is only reported for the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("sea_attn.hip", "sea_attn.hpp", "sea_attn_tile.hip", "sea_common.hpp"):
        h.update(open(os.path.join(ROOT, "sea-attention_amd", "csrc", f), "rb").read())
    return h.hexdigest()


class _Cfg:
    def __init__(self, hidden, heads, max_pos):
        self.hidden_size, self.num_attention_heads, self.max_position_embeddings = hidden, heads, max_pos


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                             # cgroup v2 quota ("max" or "<quota> <period>")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(w, n_seq=4):
    """Oracle (CPU port of the reference's dense PyTorch branch) on a bounded sample: ONE sequence of the
    workload shape through the kernel-level path (probs -> top-k -> interpolate -> dense masked attention),
    fp32, all host cores."""
    from oracle import sea_oracle as O
    cores = _usable_cores()
    torch.set_num_threads(cores)
    H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
    g = torch.Generator().manual_seed(42)
    probs = torch.softmax(torch.randn((1, H, T, T_M), generator=g), -1)
    q = torch.randn((1, H, T, d), generator=g) * d ** -0.5
    kk = torch.randn((1, H, T, d), generator=g)
    v = torch.randn((1, H, T, d), generator=g)
    rs = torch.sigmoid(torch.randn((1, H, T), generator=g))
    hc = max(1, min(H, (1 << 29) // (T * T)))            # keep the T x T temporaries around 2 GB
    t0 = time.perf_counter()
    for _ in range(n_seq):
        out, _ = O.dense_path(probs, q, kk, v, rs, k, head_chunk=hc)
        out = O.mix(out, v, torch.zeros((1, H, T)))
    dt = time.perf_counter() - t0
    return {"value": n_seq * T / dt, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n_seq} sequences x {T} tokens, kernel-level path H..K (probs given), fp32 dense branch, "
                      f"head_chunk={hc}, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="opt-1.3b", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel eagerly instead of replaying the layer as a HIP graph")
    ap.add_argument("--inspect-padding", action="store_true",
                    help="let the module inspect the mask for padding every step (reference behaviour, one host sync)")
    ap.add_argument("--cpu-seqs", type=int, default=4, help="sequences in the CPU-baseline sample")
    ap.add_argument("--prewarm", type=int, default=8,
                    help="untimed iterations before the W warm-up steps: lets MIOpen/rocBLAS settle their kernel selection")
    ap.add_argument("--kernel-iters", type=int, default=20, help="extra kernel-only iterations (H..K) after the timed steps")
    ap.add_argument("--sparse-kernel", default="auto", choices=["auto", "gather", "tile"],
                    help="kernel of steps J-L (sea_sparse_attention_ex path)")
    ap.add_argument("--no-output-check", action="store_true")
    ap.add_argument("--decode-steps", type=int, default=0,
                    help="positions of the generation leg after the timed steps (off by default: its one-row launches share kernel "
                         "names with the layer's and would dilute a profiler's per-kernel averages of the default command)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a ONE-GPU box: every rank uses cuda:0 and the process group runs on gloo -- exercises the "
                         "multi-rank code path of this script (shards, graph capture beside a process group, pipelined gather, "
                         "max-over-ranks timing); its numbers mean nothing")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if args.rehearse else "nccl", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs the MI355X"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    import sea_attention_amd as S
    from sea_attention_amd import _lib, distributed as D
    from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention, ops
    _lib.load(build_if_missing=True)

    w = WORKLOADS[args.workload]
    H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    NB = args.batch

    S.seed(42)
    pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=w["nbf"], causal=True,
                               k_flatten=True, k_flatten_dim='causal_batch', context_output_method='mix')
    layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc).to(dev).to(dtype).eval()
    for m in layer.modules():
        if hasattr(m, 'benchmarking'):
            m.benchmarking = True
    layer.attention.context_layer_dtype = dtype          # the consumer (out_proj) runs in `dtype`
    # the synthetic batch carries no padding by construction; telling the module removes the per-layer host
    # sync the reference pays to find that out (attention.py:434) and lets the CPU enqueue ahead of the GPU
    layer.attention.assume_not_padded = None if args.inspect_padding else True   # None: the module inspects the mask itself
    layer.attention.sparse_kernel = args.sparse_kernel
    torch.manual_seed(42 + rank)
    q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(dtype)
    kk = torch.randn((NB, H, T, d), device=dev).to(dtype)
    v = torch.randn((NB, H, T, d), device=dev).to(dtype)
    fp_min = torch.finfo(torch.float16 if dtype != torch.float32 else torch.float32).min / 2
    mask = ((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min)
    mask = mask.view(1, 1, T, T).to(dtype).expand(NB, 1, T, T).contiguous()

    bench = S.get_bench()

    # N > 1: the all-gather of step i (RCCL, its own stream) overlaps the compute of step i+1 -- two slots of
    # (local shard, gathered output); every collective is awaited before the timed region ends (gather.finish()).
    gather = D.ContextGatherer((NB, T, H * d), NB * world, dtype, dev) if world > 1 else None

    def step():
        with torch.no_grad():
            out = layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
            ctx = out.context_layer
            if world > 1:                                  # eager mode only: the graph path writes the slot directly
                slot = gather.next_slot()
                gather.local[slot].copy_(ctx)
                ctx = gather.launch(slot)
        return out, ctx

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.prewarm + args.warmup):
        step()
    graph = None
    attn_events = []
    if not args.eager:
        # HIP-graph mode (default).  Everything of the layer step UP TO the fused sparse-attention launch is
        # captured into one graph (torch ops and the C-ABI kernels alike go to the capturing stream); the
        # attention kernel itself -- the last launch of the layer, which writes context_layer -- is launched
        # eagerly right after each replay, bracketed by HIP events on the same stream (roofline timing).
        from sea_attention_amd.perlin_attention import attention as _A
        real_attn = _A.ops.sparse_attention
        rec = {}

        def recorder(*a, **kw):
            rec["a"], rec["kw"] = a, kw
            return kw.get("out")
        try:
            if gather is not None:
                gather.finish()                             # no collective in flight while the capture runs
            sync_all()
            _A.ops.sparse_attention = recorder
            graph = torch.cuda.CUDAGraph()
            # with a process group alive its watchdog thread polls events: keep the capture's error mode thread-local
            with torch.cuda.graph(graph, **({"capture_error_mode": "thread_local"} if world > 1 else {})):
                with torch.no_grad():
                    g_out = layer(None, None, None, query_layer=q, key_layer=kk, value_layer=v, attention_mask=mask)
        except Exception as e:                                  # capture unsupported on this stack: stay eager
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
            graph = None
        finally:
            _A.ops.sparse_attention = real_attn
        if graph is not None and "a" in rec:
            eager_step = step

            def step():
                kw = rec["kw"]
                if world > 1:      # the attention kernel writes straight into this step's gather slot
                    slot = gather.next_slot()
                    kw = dict(kw, out=gather.local[slot].view(NB, T, H, d).permute(0, 2, 1, 3))
                graph.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                real_attn(*rec["a"], **kw)
                e1.record()
                attn_events.append((e0, e1))
                ctx = gather.launch(slot) if world > 1 else g_out.context_layer
                return g_out, ctx
            for _ in range(3):
                step()
            attn_events.clear()
        else:
            graph = None
    # per-kernel HIP events through the module's named regions (events only, no host sync inside the steps)
    bench.disabled, bench.synchronize = (graph is not None), True   # regions cannot be timed inside a graph replay
    bench.reset_measures()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, ctx = step()
    t_enqueued = time.perf_counter() - t0                 # host time to enqueue all K steps (no sync inside)
    if gather is not None:
        gather.finish()                                   # every step's all-gather completes inside the timed region
    sync_all()
    elapsed = time.perf_counter() - t0
    regions = bench.todict()                              # seconds per call, from HIP events on the launch stream
    t_attn_graph = None
    if graph is not None:
        t_attn_graph = sum(a.elapsed_time(b) for a, b in attn_events) / max(1, len(attn_events)) / 1e3
        bench.disabled = False                            # informational per-region times: a few eager steps
        for _ in range(5):
            eager_step()
        torch.cuda.synchronize()
        regions = bench.todict()
    bench.disabled, bench.synchronize = True, False
    bench.reset_measures()

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    tokens_per_s = NB * world * T / (elapsed / args.steps)

    # ---- output self-check (outside the timed region): the batch the bench times is also a batch that is RIGHT -------
    # item n of the batched output == the same item run ALONE through the whole layer, bit for bit (every kernel is
    # deterministic and treats batch items independently; the Performer is told to take the one-pass kernel the batch
    # takes instead of cutting the lone sequence into segments), and its CSR is rebuilt from its map by the unfused
    # top-k path as a second opinion
    output_check = None
    if not args.no_output_check:
        with torch.no_grad():
            # N > 1 in graph mode: the attention launch wrote this rank's shard straight into the gathered buffer
            ctx_b = ctx[rank * NB:(rank + 1) * NB] if world > 1 else out.context_layer
            probs_b = out.estimated_attention_probs_m
            keep_t = ops.keep_table_causal(H, T, T_M, k, device=dev)
            zc = ops.z_capacity(keep_t.cpu(), H, T, T, T_M, k, True)
            bits_ok, worst = True, 0.0
            # the lone item takes the Performer path the BATCH took (one pass for a full batch; for a one-sequence workload
            # the library's sequence-parallel plan of that very shape)
            layer.attention.performer_segments = ops.performer_plan(NB, H, T, d, layer.attention.performer_nb_features, dtype)[0]
            for n_ in sorted({0, NB - 1}):
                csr_n, _ = ops.topk_to_csr(probs_b[n_:n_ + 1].contiguous(), keep_t, k, target_width=T, z_cap=zc)
                csr_b = out.partial_attention_mask
                z_ = int(csr_n.crow[0, -1].item())
                bits_ok &= bool(torch.equal(csr_n.crow[0], csr_b.crow[n_]) and torch.equal(csr_n.col[0, :z_], csr_b.col[n_, :z_]))
                alone = layer(None, None, None, query_layer=q[n_:n_ + 1], key_layer=kk[n_:n_ + 1], value_layer=v[n_:n_ + 1],
                              attention_mask=mask[n_:n_ + 1])
                same_map = bool(torch.equal(alone.estimated_attention_probs_m, probs_b[n_:n_ + 1]))
                d_ = (alone.context_layer.float() - ctx_b[n_:n_ + 1].float())
                worst = max(worst, (d_.norm() / ctx_b[n_:n_ + 1].float().norm()).item())
                bits_ok &= same_map and bool(torch.equal(alone.context_layer, ctx_b[n_:n_ + 1]))
            layer.attention.performer_segments = None
            finite = bool(torch.isfinite(ctx_b.float()).all().item())
            gathered_ok = None
            if world > 1:
                # the all-gather put every rank's shard where it belongs: each rank's checksum of its own shard, exchanged,
                # against the checksum of the matching chunk of the gathered buffer this rank holds
                mine = ctx_b.float().abs().sum(dtype=torch.float64).view(1)
                sums = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(sums, mine)
                here = [ctx[r * NB:(r + 1) * NB].float().abs().sum(dtype=torch.float64).view(1) for r in range(world)]
                gathered_ok = all(bool(torch.equal(a, b)) for a, b in zip(sums, here))
            output_check = {"status": "ok" if (bits_ok and finite and gathered_ok is not False) else "FAILED",
                            "items_alone_bitwise_equal_to_batched_rows": bits_ok, "finite": finite,
                            "items_checked": sorted({0, NB - 1}), "layer_item_alone_rel_diff": round(worst, 8)}
            if gathered_ok is not None:
                output_check["gathered_shards_match_their_ranks"] = gathered_ok

    # ---- roofline of the dominant HIP kernel (fused sparse attention) ---------------------------------
    # achieved = SURVEY 8d's algorithmic bytes (every gathered K / V row counted once per entry) / launch time.  The
    # K + V of one head (1 MiB) stay in the XCD's L2, so what binds is the L2 -> L1 request path, not HBM: `bound` says so,
    # `frac` is still against the 8 TB/s HBM line (the contract's roofline), `l2_gather_frac` against the 17.8 TB/s the
    # microarchitecture guide measures for L2-served row gathers, `frac_compulsory_hbm` = bytes that MUST cross HBM once
    # (q, k, v, avg, out, col, offsets, scales) / time / 8 TB/s.
    csr = out.partial_attention_mask
    Z = int(csr.crow[:, -1].sum().item())
    esz = torch.tensor([], dtype=dtype).element_size()
    alg_bytes = ops.sparse_attention_bytes(Z, NB, H, T, d, esz)
    compulsory = (5 * NB * H * T * d * esz                      # q, k, v, avg in; out
                  + Z * 4 + NB * T * (H + 2) * 4                # col, head_off, crow
                  + 2 * NB * H * T * 4)                         # row_scale, mix
    t_attn = t_attn_graph if t_attn_graph else regions.get('attention.sparse.fused')
    roof = None
    if t_attn:
        achieved = alg_bytes / t_attn / 1e9
        traffic, traffic_note = None, "no PMC pass on record (scripts/gpu_pmc.sh writes profiles/traffic_latest.json)"
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            try:
                rec = json.load(open(tp))
                if rec.get("kernel_source_sha256") == _attn_source_sha() and rec.get("nnz") != Z:
                    traffic_note = "profiles/traffic_latest.json was taken on another workload (entry count differs): not reported"
                elif rec.get("kernel_source_sha256") == _attn_source_sha():
                    traffic = rec.get("sea_sparse_attention_hbm_bytes_per_launch")
                    traffic_note = rec.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command")
                else:
                    traffic_note = "profiles/traffic_latest.json was taken on other kernel sources: not reported"
            except Exception:
                pass
        gname = "sparse_attn_rows80_kernel" if d == 80 and esz == 2 else "sparse_attn_rows_kernel"
        kname = {"tile": "sparse_attn_tile_kernel", "gather": gname,
                 "auto": gname + " + sparse_attn_tile_kernel (per-block dispatch: both launches inside the timed events)"}[args.sparse_kernel]
        roof = {"bound": "l2_gather", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                "compulsory_bytes": compulsory, "frac_compulsory_hbm": round(compulsory / t_attn / 1e9 / HBM_PEAK_GBS, 4),
                "l2_gather_peak": L2_GATHER_GBS, "l2_gather_frac": round(achieved / L2_GATHER_GBS, 4),
                "timing": "HIP events around every launch inside the timed steps" if graph is not None
                          else "HIP events of the module's 'attention.sparse.fused' region inside the timed steps",
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(t_attn * 1e3, 4), "nnz": Z}

    # ---- kernel-level path only (H..K on HIP, probs given) -- what the CPU baseline below also runs -------
    kernel_path = None
    if args.kernel_iters > 0:
        probs = torch.softmax(torch.randn((NB, H, T, T_M), device=dev), -1).to(dtype)
        keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
        z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)
        rs = torch.sigmoid(torch.randn((NB, H, T), device=dev))
        mx = torch.sigmoid(torch.randn((NB, H, T), device=dev))
        avg = (v.float().cumsum(-2) / torch.arange(1, T + 1, device=dev).view(1, 1, -1, 1)).to(dtype)
        ctx2 = torch.empty((NB, T, H * d), dtype=dtype, device=dev)

        def kstep():
            c, _ = ops.topk_to_csr(probs, keep, k, target_width=T, z_cap=z_cap)
            ops.sparse_attention(q, kk, v, c, row_scale=rs, avg=avg, mix=mx, out=ctx2.view(NB, T, H, d).permute(0, 2, 1, 3), path=args.sparse_kernel)
            return c
        for _ in range(3):
            kstep()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        tk = ta = 0.0
        for _ in range(args.kernel_iters):
            e0.record(); c, _ = ops.topk_to_csr(probs, keep, k, target_width=T, z_cap=z_cap); e1.record()
            ops.sparse_attention(q, kk, v, c, row_scale=rs, avg=avg, mix=mx, out=ctx2.view(NB, T, H, d).permute(0, 2, 1, 3), path=args.sparse_kernel)
            e2.record(); torch.cuda.synchronize()
            tk += e0.elapsed_time(e1); ta += e1.elapsed_time(e2)
        tk /= args.kernel_iters; ta /= args.kernel_iters
        Zk = int(c.crow[:, -1].sum().item())
        kb = ops.sparse_attention_bytes(Zk, NB, H, T, d, esz)
        kernel_path = {"value": round(NB * T / ((tk + ta) / 1e3), 1), "unit": "tokens/s",
                       "topk_interp_csr_ms": round(tk, 4), "sparse_attention_ms": round(ta, 4),
                       "sparse_attention_GBs": round(kb / (ta / 1e3) / 1e9, 1), "nnz": Zk,
                       "note": "HIP kernels only (steps H..K), softmax(randn) probability map, per GPU"}
        # the same leg on a map whose neighbouring rows share their keys (what a trained predictor emits: diagonal band,
        # vertical stripes, sink -- sea_attention_amd/synthetic.py) with the per-block dispatch plan, as the layer runs it
        if dtype != torch.float32 and d in (64, 80, 128):
            from sea_attention_amd import synthetic
            sp = synthetic.structured_probs(NB, H, T, T_M, dev, dtype, seed=1)
            cs, _ = ops.topk_to_csr(sp, keep, k, target_width=T, z_cap=z_cap)
            del sp
            times = {}
            for name_, kw_ in (("gather", dict(path="gather")), ("tile", dict(path="tile")), ("auto", dict(path="auto"))):
                for it_ in range(2 + args.kernel_iters):
                    if it_ == 2:
                        torch.cuda.synchronize(); e0.record()
                    pl_ = ops.attention_plan(cs, T_M) if name_ == "auto" else None
                    ops.sparse_attention(q, kk, v, cs, row_scale=rs, avg=avg, mix=mx, out=ctx2.view(NB, T, H, d).permute(0, 2, 1, 3),
                                         plan=pl_, **kw_)
                e1.record(); torch.cuda.synchronize()
                times[name_] = round(e0.elapsed_time(e1) / args.kernel_iters, 4)
            Zs = int(cs.crow[:, -1].sum().item())
            kernel_path["structured_map"] = {"sparse_attention_ms": times, "nnz": Zs,
                                             "algorithmic_GBs_auto": round(ops.sparse_attention_bytes(Zs, NB, H, T, d, esz) / (times["auto"] / 1e3) / 1e9, 1),
                                             "note": "auto = plan kernel + both gated launches (what the layer runs)"}

    # ---- generation leg: one position per step from a (T - 64)-token prefix, the step replayed as a HIP graph
    decode = None
    if args.decode_steps > 0 and world == 1 and dtype != torch.float32:
        try:
            import copy
            from sea_attention_amd.perlin_attention.decode import DecodeSession
            lc_ = copy.deepcopy(layer)
            lc_.pconfig = copy.copy(layer.pconfig); lc_.pconfig.use_cache = True
            lc_.attention.pconfig = lc_.pconfig
            T0 = T - 64
            with torch.no_grad():
                pre = lc_(None, None, None, query_layer=q[:, :, :T0], key_layer=kk[:, :, :T0], value_layer=v[:, :, :T0],
                          attention_mask=mask[:, :, :T0, :T0].contiguous())
                sess = DecodeSession(lc_.attention, pre.state, kk[:, :, :T0], v[:, :, :T0], capacity=T, use_graph=True)
                nd = min(args.decode_steps, 60)
                for i in range(4):
                    sess.step(q[:, :, T0 + i:T0 + i + 1], kk[:, :, T0 + i:T0 + i + 1], v[:, :, T0 + i:T0 + i + 1])
                torch.cuda.synchronize(); t0_ = time.perf_counter()
                for i in range(4, 4 + nd):
                    sess.step(q[:, :, T0 + i:T0 + i + 1], kk[:, :, T0 + i:T0 + i + 1], v[:, :, T0 + i:T0 + i + 1])
                torch.cuda.synchronize()
                t_pos = (time.perf_counter() - t0_) / nd
            decode = {"ms_per_position": round(t_pos * 1e3, 4), "tokens_per_s": round(NB / t_pos, 1), "prefix_tokens": T0,
                      "positions_timed": nd, "note": "DecodeSession: fixed-capacity caches, position in device memory, "
                      "the step's launches replayed as one HIP graph; all sequences of the batch advance together"}
            del sess, lc_, pre
        except Exception as e:                                   # an extra leg never takes the headline line down
            decode = {"error": f"{type(e).__name__}: {e}"[:200]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, args.cpu_seqs)
        # like-for-like twin of `value` (the WHOLE layer, steps A..L): the reference's own CPU-runnable form of this layer is
        # its dense torch mode (BASELINE config 0 runs it on CPU); same weights, one sequence, fp32, all host cores
        try:
            import copy
            cores_ = _usable_cores()
            torch.set_num_threads(cores_)
            lc = copy.deepcopy(layer).to("cpu").float().eval()
            for m in lc.modules():
                if hasattr(m, 'benchmarking'):
                    m.benchmarking = False
            q1, k1, v1 = (t_[:1].float().cpu() for t_ in (q, kk, v))
            m1 = mask[:1].float().cpu()
            with torch.no_grad():
                tc0 = time.perf_counter()
                lc(None, None, None, query_layer=q1, key_layer=k1, value_layer=v1, attention_mask=m1)
                dtc = time.perf_counter() - tc0
            cpu["full_layer"] = {"value": round(T / dtc, 1), "unit": "tokens/s", "cores": cores_, "kind": "port",
                                 "sample": f"1 sequence x {T} tokens through the whole layer (steps A..L) in the dense torch mode "
                                           f"(the reference's CPU-runnable path), fp32, {dtc:.1f} s"}
            del lc
        except Exception as e_:                                   # the twin is informational: never fail the bench line on it
            cpu["full_layer"] = {"error": f"{type(e_).__name__}: {e_}"[:200]}

    if rank == 0:
        line = {
            "metric": "tokens/sec + achieved HBM GB/s, OPT-1.3B SEA T=4096 k=64, 1/2/4/8 MI355X",
            "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload} SEA attention layer forward (sparse mode, steps A-L), "
                                   f"H={H} d={d} T={T} k={k} predictor_length={T_M} nbf={w['nbf']}, "
                                   f"batch {NB} sequences/GPU, random-init weights seed 42, "
                                   f"context_layer {args.dtype} (reference default: fp32), "
                                   + ("unpadded batch declared to the module (assume_not_padded: no mask inspection sync)"
                                      if not args.inspect_padding else "module inspects the mask for padding (one host sync)")
                                   + f", sparse kernel path {args.sparse_kernel}"
                                   + (", + RCCL all-gather of context shards" if world > 1 else "")
                                   + (", layer replayed as a HIP graph + eager fused-attention launch" if graph is not None
                                      else ", eager launches"),
                       "global_batch": NB * world, "seq_len": T, "parallelism": f"dp{world} (batch shards)"},
            "roofline": roof, "cpu_baseline": cpu, "output_check": output_check, "kernel_path": kernel_path,
            "decode": decode, "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 3),
            "regions_ms": {k_: round(v_ * 1e3, 4) for k_, v_ in sorted(regions.items())},
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
