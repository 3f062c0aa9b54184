#!/usr/bin/env python3
"""bench.py -- SEA sparse-attention layer on MI355X: tokens/s + achieved HBM GB/s of the sparse kernel.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  One JSON line on rank 0.
  * N = 1: one process, one GPU.
  * N > 1 with WORLD_SIZE set (the driver's `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`):
    this process is one of the N ranks (one rank per GPU, RCCL).  Every rank checks WORLD_SIZE == --gpus and exits
    non-zero when they disagree: the line can never claim more GPUs than ran.
  * N > 1 with no WORLD_SIZE: the script launches its own N ranks (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`) as a child process BEFORE anything
    in this process touches the GPU, relays the child's output and exits with its return code.

A "step" = one forward of ONE SEA attention layer in sparse mode (`benchmarking=True`, steps A..L of SURVEY.md
section 3B: value augmentation, Performer, predictor MLP+CNN, softmax, HIP grouped top-k + interpolation -> flat CSR,
HIP fused sparse attention + mix) on a synthetic batch, followed -- when N>1 -- by the RCCL all-gather of the context
shards.  Workload = BASELINE.json configs[2]: OPT-1.3B SEA, H=32 d=64 T=4096, k=64, predictor length 256,
nb-factor 8, bf16, random-init weights (seed 42).  Per-GPU batch is fixed (weak scaling).  The other BASELINE shapes
(configs[1], [3], [4]) run as short legs after the headline and are reported under `other_workloads`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "opt-1.3b": dict(H=32, d=64, T=4096, T_M=256, k=64, nbf=8),
    "opt-125m": dict(H=12, d=64, T=2048, T_M=256, k=64, nbf=8),
    "opt-2.7b": dict(H=32, d=80, T=8192, T_M=256, k=64, nbf=8),
    "llama-13b": dict(H=40, d=128, T=4096, T_M=256, k=64, nbf=8),
}
# short legs after the headline: (workload, sequences per GPU) -- BASELINE.json configs[1], [3], [4] at the per-GPU batch
# their configuration puts on one MI355X
OTHER_LEGS = (("opt-125m", 8), ("opt-2.7b", 1), ("llama-13b", 1))
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_GATHER_GBS = 17800.0    # MI355X_MICROARCH.md "Indexed rows": 16.8-18.8 TB/s chip-wide for rows served by the XCD's L2
L2_PEAK_GBS = 34500.0      # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate streaming rate of the eight L2s
METRIC = "tokens/sec + achieved HBM GB/s, OPT-1.3B SEA T=4096 k=64, 1/2/4/8 MI355X"


def parse_args(argv=None):
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
                    help="untimed iterations before the W warm-up steps: lets the runtime settle")
    ap.add_argument("--kernel-iters", type=int, default=20, help="extra kernel-only iterations (H..K) after the timed steps")
    ap.add_argument("--sparse-kernel", default="ab", choices=["ab", "auto", "gather", "tile"],
                    help="kernel of steps J-L (sea_sparse_attention_ex path); 'ab' times the candidates on the layer's own "
                         "selection before the timed region and runs the fastest (reported as attention_path_ab)")
    ap.add_argument("--no-output-check", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short legs over the other BASELINE shapes")
    ap.add_argument("--other-steps", type=int, default=20, help="timed steps of each short leg")
    ap.add_argument("--context-dtype", default="auto", choices=["auto", "fp32", "same"],
                    help="dtype of context_layer: fp32 = the module's and the reference's default (flat_csr_sdbmm.py:347); "
                         "'same' = the data dtype (reported as the context_bf16 leg by the default run); auto = fp32 on one GPU, "
                         "the data dtype for N > 1 (the context shard is what the all-gather carries over xGMI: 134 MB instead "
                         "of 268 MB per rank and step at the headline shape)")
    ap.add_argument("--no-grid", action="store_true",
                    help="skip the reference's own ablation grid (benchmark_opt_ablation.py:160-186: opt-125m layer, batch 1, T = 2048, "
                         "k in {32, 64, 128} x predictor length in {64, 128, 256, 384}, + exp_long_context.py:152's T_M = 96 / k = 128)")
    ap.add_argument("--gather-chunks", type=int, default=1,
                    help="N > 1: cut every step's context shard into this many groups of sequences, each written by its own "
                         "attention launch and sent by its own in-place all-gather as soon as that launch is enqueued "
                         "(distributed.ChunkedContextGatherer); 1 = one launch, one collective per step")
    ap.add_argument("--no-long-context", action="store_true", help="skip the 32768-token leg (opt-125m shape, one sequence)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32-DATA leg at BASELINE config 2 (B=8 H=12 T=2048 d=64)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the forward + backward leg of the sparse branch")
    ap.add_argument("--decode-steps", type=int, default=200,
                    help="positions of the generation leg (batch 1 and the headline batch) after the timed steps; 0 = off (the "
                         "profiling scripts pass 0: its one-row launches share kernel names with the layer's and would dilute a "
                         "profiler's per-kernel averages)")
    ap.add_argument("--seq-len", type=int, default=0,
                    help="override the workload's sequence length T (profiling the long-context shape: --workload opt-125m --batch 1 "
                         "--seq-len 32768); 0 = the workload's own")
    ap.add_argument("--repeats", type=int, default=4,
                    help="after the timed K steps, time K steps this many more times and report min / median / max of ms_per_step "
                         "(`repeats`); `value` stays the first timed region")
    ap.add_argument("--no-eager-outputs", action="store_true",
                    help="skip the leg that materialises estimated_attention_probs and the CSR's col_indices every step "
                         "(the reference's eager outputs, attention.py:1343 + causal_resize_m_to_t.py:757-762)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a box without N GPUs: the process group runs on gloo; with one GPU every rank uses cuda:0 and "
                         "walks the multi-rank code path of this script (shards, graph capture beside a process group, pipelined "
                         "gather, max-over-ranks timing); with NO GPU a CPU stand-in step exercises launcher, process group, "
                         "gather and timing protocol only.  Its numbers mean nothing")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """--gpus N > 1 outside a launcher: start N ranks of this script under torch.distributed.run and relay them.
    Nothing in THIS process has touched the GPU (no HIP call, no torch.cuda.* besides none at all)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for out in proc.stdout:                                    # relay; the JSON line is rank 0's last '{' line
        sys.stdout.write(out)
        sys.stdout.flush()
        if out.lstrip().startswith("{"):
            line = out
    rc = proc.wait()
    if rc == 0 and line is None:
        print("[bench] the ranks exited 0 but printed no result line", file=sys.stderr)
        rc = 3
    return rc


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


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


def _attn_source_sha():
    """sha256 over the sources of the graded kernel: a PMC traffic record is only reported for the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("sea_attn.hip", "sea_attn.hpp", "sea_attn_tile.hip", "sea_common.hpp"):
        h.update(open(os.path.join(ROOT, "sea-attention_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def cpu_baseline(w, n_seq=4):
    """Oracle (CPU port of the reference's dense PyTorch branch) on a bounded sample: ONE sequence of the
    workload shape through the kernel-level path (probs -> top-k -> interpolate -> dense masked attention),
    fp32, all host cores."""
    import torch
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
    return {"value": n_seq * T / dt, "unit": "tokens/s", "cores": cores, "cpu_model": _cpu_model(), "kind": "port",
            "sample": f"{n_seq} sequences x {T} tokens, kernel-level path H..K (probs given), fp32 dense branch, "
                      f"head_chunk={hc}, {dt:.1f} s"}


# ---------------------------------------------------------------------------------------------------------------------
class LayerBench:
    """One SEA attention layer of a BASELINE shape with its synthetic batch on this rank's GPU: eager step, HIP-graph
    step (everything up to the fused sparse-attention launch captured, that launch eager between HIP events), the
    A/B of the attention kernel paths, and the roofline block of the attention launch."""

    def __init__(self, wname, NB, dtype_name, dev, ctx_dtype_name=None, inspect_padding=False, seed_offset=0, override=None,
                 layer_attrs=None):
        import torch
        import sea_attention_amd as S
        from sea_attention_amd.perlin_attention import PerlinAttentionConfig, PerlinSelfAttention
        self.torch, self.S = torch, S
        self.wname, self.w, self.NB, self.dev = wname, dict(WORKLOADS[wname], **(override or {})), NB, dev
        w = self.w
        H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
        dts = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}
        self.dtype = dts[dtype_name]
        self.ctx_dtype = dts[ctx_dtype_name or "fp32"]            # fp32 = the module's default (context_layer_dtype None)
        S.seed(42)
        pc = PerlinAttentionConfig(k=k, attention_predictor_length=T_M, performer_nb_factor=w["nbf"], causal=True,
                                   k_flatten=True, k_flatten_dim='causal_batch', context_output_method='mix')
        # the random-init weights are built on the CPU, and the Performer's projection goes through a QR whose rounding follows
        # the thread count: built single-threaded, so that every leg of every run (with or without the CPU-baseline leg, which
        # sets the thread count) times the SAME layer -- a leg's entry count is what matches it to its PMC record
        nthr = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            layer = PerlinSelfAttention(_Cfg(H * d, H, T), pc)
        finally:
            torch.set_num_threads(nthr)
        layer = layer.to(dev).to(self.dtype).eval()
        for m in layer.modules():
            if hasattr(m, 'benchmarking'):
                m.benchmarking = True
        layer.attention.context_layer_dtype = None if self.ctx_dtype == torch.float32 else self.ctx_dtype   # None = the default
        # the synthetic batch carries no padding by construction; telling the module removes the per-layer host
        # sync the reference pays to find that out (attention.py:434) and lets the CPU enqueue ahead of the GPU
        layer.attention.assume_not_padded = None if inspect_padding else True
        for a_, v_ in (layer_attrs or {}).items():              # e.g. the eager-outputs twin: lazy_attention_probs / lazy_csr_columns off
            assert hasattr(layer.attention, a_), a_
            setattr(layer.attention, a_, v_)
        self.layer = layer
        torch.manual_seed(42 + seed_offset)
        self.q = (torch.randn((NB, H, T, d), device=dev) * d ** -0.5).to(self.dtype)
        self.k = torch.randn((NB, H, T, d), device=dev).to(self.dtype)
        self.v = torch.randn((NB, H, T, d), device=dev).to(self.dtype)
        fp_min = torch.finfo(torch.float16 if self.dtype != torch.float32 else torch.float32).min / 2
        mask = ((torch.arange(T, device=dev).view(1, T) > torch.arange(T, device=dev).view(T, 1)) * fp_min)
        self.mask = mask.view(1, 1, T, T).to(self.dtype).expand(NB, 1, T, T).contiguous()
        self.graph, self.rec, self.g_out = None, None, None
        self._pending = None
        self.attn_events = []
        self.out_override = None          # N > 1: callable(step) -> the (N,H,T,d)-shaped view the attention launch writes
        self.capture_error = None

    # -- one forward, eager ---------------------------------------------------------------------------------------------
    def forward(self):
        with self.torch.no_grad():
            return self.layer(None, None, None, query_layer=self.q, key_layer=self.k, value_layer=self.v,
                              attention_mask=self.mask)

    # -- HIP-graph form ---------------------------------------------------------------------------------------------------
    def capture(self, sparse_kernel, with_process_group=False):
        """Capture everything of the layer step UP TO the fused sparse-attention launch (torch ops and the C-ABI kernels
        alike go to the capturing stream); the attention launch -- the last of the layer, it writes context_layer -- is
        recorded and launched eagerly after each replay between HIP events on the same stream (roofline timing)."""
        torch = self.torch
        from sea_attention_amd.perlin_attention import attention as _A
        self.layer.attention.sparse_kernel = sparse_kernel
        self.graph, self.rec, self.g_out = None, None, None
        real_attn = _A.ops.sparse_attention
        rec = {}

        def recorder(*a, **kw):
            rec["a"], rec["kw"] = a, kw
            return kw.get("out")
        try:
            torch.cuda.synchronize()
            _A.ops.sparse_attention = recorder
            graph = torch.cuda.CUDAGraph()
            # with a process group alive its watchdog thread polls events: keep the capture's error mode thread-local
            with torch.cuda.graph(graph, **({"capture_error_mode": "thread_local"} if with_process_group else {})):
                g_out = self.forward()
        except Exception as e:                                  # capture unsupported on this stack: stay eager
            self.capture_error = f"{type(e).__name__}: {e}"[:300]
            print(f"[bench] HIP-graph capture failed ({self.capture_error}); falling back to eager launches", file=sys.stderr)
            return False
        finally:
            _A.ops.sparse_attention = real_attn
        if "a" not in rec:
            self.capture_error = "the layer issued no fused sparse-attention launch"
            return False
        self.graph, self.rec, self.g_out, self._real_attn = graph, rec, g_out, real_attn
        # sparse_kernel = "gather": the layer leaves the CSR's column array to the attention launch (steps I + J fused), and
        # the handle says so; a replayed step selects afresh, so the handle is put back into that state before every launch
        self._pending = rec["a"][3]._pending
        return True

    def step(self, out_view=None, chunk_views=None, after_chunk=None):
        """One step.  Graph mode: replay + the eager attention launch (optionally redirected into `out_view`, the rank's
        slot of the gathered buffer); returns (layer output tuple, context tensor written)."""
        torch = self.torch
        if self.graph is None:
            out = self.forward()
            return out, out.context_layer
        kw = self.rec["kw"] if out_view is None else dict(self.rec["kw"], out=out_view)
        self.graph.replay()
        if self._pending is not None:
            self.rec["a"][3]._pending = self._pending
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if chunk_views is None:
            self._real_attn(*self.rec["a"], **kw)
        else:
            # one attention launch per group of sequences (FlatCSR.items: the same arrays, sliced), each writing its chunk of
            # the gather slot; `after_chunk(c)` enqueues that chunk's collective behind it
            q_, k_, v_, csr_ = self.rec["a"][:4]
            n_c = self.NB // len(chunk_views)
            cut = lambda t_, a_, b_: t_[a_:b_] if t_ is not None else None
            for c, view in enumerate(chunk_views):
                a_, b_ = c * n_c, (c + 1) * n_c
                kw_c = dict(kw, out=view, row_scale=cut(kw.get("row_scale"), a_, b_), avg=cut(kw.get("avg"), a_, b_),
                            mix=cut(kw.get("mix"), a_, b_))
                self._real_attn(q_[a_:b_], k_[a_:b_], v_[a_:b_], csr_.items(a_, b_), *self.rec["a"][4:], **kw_c)
                if after_chunk is not None:
                    after_chunk(c)
        e1.record()
        self.attn_events.append((e0, e1))
        return self.g_out, (self.g_out.context_layer if (out_view is None and chunk_views is None) else None)

    def attn_ms(self):
        if not self.attn_events:
            return None
        return sum(a.elapsed_time(b) for a, b in self.attn_events) / len(self.attn_events)

    def timed(self, steps):
        """K steps between two device synchronisations; returns (seconds, mean attention-launch ms or None)."""
        torch = self.torch
        self.attn_events.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, self.attn_ms()

    def tile_supported(self):
        return self.dtype != self.torch.float32 and self.w["d"] in (64, 80, 128)

    def choose_path(self, requested, with_process_group=False, steps=4):
        """'ab': capture the step with each candidate path of the attention launch, time `steps` replays of each on the
        layer's own selection, keep the fastest WHOLE STEP (the plan launch of 'auto' sits inside the graph).  Returns
        (path, report)."""
        if requested != "ab":
            ok = self.capture(requested, with_process_group)
            return requested, {"requested": requested, "graph": ok}
        cands = ["gather"] + (["auto", "tile"] if self.tile_supported() else [])
        rep = {}
        for rnd in range(2):                                   # two interleaved rounds, the better of each candidate: whoever
            for c in (cands if rnd == 0 else cands[::-1]):     # is measured first pays the process's one-time costs, and a
                # candidate measured right after another one's capture inherits its cache state: second round in reverse order
                if not self.capture(c, with_process_group):
                    return "auto" if self.tile_supported() else "gather", {"requested": "ab", "graph": False}
                self.timed(2)
                dt, am = self.timed(steps)
                cur = {"ms_per_step": round(dt / steps * 1e3, 4), "attention_ms": round(am, 4)}
                if c not in rep or cur["ms_per_step"] < rep[c]["ms_per_step"]:
                    rep[c] = cur
        best = min(rep, key=lambda c_: rep[c_]["ms_per_step"])
        self.capture(best, with_process_group)
        return best, {"requested": "ab", "graph": True, "candidates": rep, "chosen": best,
                      "note": f"2 x {steps} graph-replayed steps per candidate (two rounds, the second in reverse order; the better round counts) on the "
                              "layer's own selection, before the timed region"}

    # -- roofline of the attention launch -----------------------------------------------------------------------------------
    def roofline(self, out, t_attn_s, path, timing_note):
        torch = self.torch
        from sea_attention_amd.perlin_attention import ops
        w, NB = self.w, self.NB
        H, d, T = w["H"], w["d"], w["T"]
        csr = out.partial_attention_mask
        Z = int(csr.crow[:, -1].sum().item())
        esz = torch.tensor([], dtype=self.dtype).element_size()
        alg_bytes = ops.sparse_attention_bytes(Z, NB, H, T, d, esz)
        osz = torch.tensor([], dtype=self.ctx_dtype).element_size()
        fused_ij_c = path in ("gather", "auto") and ops.fused_interp_supported(self.dtype, d, w["T_M"])
        compulsory = (4 * NB * H * T * d * esz + NB * H * T * d * osz     # q, k, v, avg in; out
                      + (0 if getattr(self.layer.attention, "lazy_csr_columns", False) and fused_ij_c else Z * 4)   # col: read, or written by
                      + NB * T * (H + 2) * 4                              # the fused form unless left pending; head_off, crow
                      + 2 * NB * H * T * 4)                               # row_scale, mix
        if not t_attn_s:
            return None, Z
        achieved = alg_bytes / t_attn_s / 1e9
        fused_ij = path in ("gather", "auto") and ops.fused_interp_supported(self.dtype, d, w["T_M"])
        same_launch = lambda a_, b_: a_ == b_ or (fused_ij and {a_, b_} <= {"gather", "auto"})   # both run the fused gather launch
        traffic, l2_req, traffic_note = None, None, "no PMC pass on record (scripts/gpu_pmc.sh + scripts/pmc_to_traffic.py write profiles/traffic_*.json)"
        # one record per workload (profiles/traffic_<workload>.json, traffic_latest.json = the headline's): the counters belong to
        # the launch they were taken on -- same kernel sources, same entry count, same kernel path, same output dtype
        import glob
        why = []
        for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json"))):
            try:
                rec = json.load(open(tp))
            except Exception:
                continue
            name = os.path.basename(tp)
            if rec.get("kernel_source_sha256") != _attn_source_sha():
                why.append(f"{name}: other kernel sources")
            elif rec.get("nnz") != Z:
                why.append(f"{name}: another workload (entry count differs)")
            elif (rec.get("attention_path") is not None and not same_launch(rec.get("attention_path"), path)) \
                    or rec.get("context_dtype") not in (None, str(self.ctx_dtype)):
                why.append(f"{name}: path {rec.get('attention_path')} / context {rec.get('context_dtype')}, this run is {path} / {self.ctx_dtype}")
            else:
                traffic = rec.get("sea_sparse_attention_hbm_bytes_per_launch")
                l2_req = rec.get("sea_sparse_attention_l2_requests_per_launch")
                if l2_req is None:                                   # records of round 3: sum over the attention kernels
                    l2_req = sum(v_.get("l2_requests_per_launch", 0) for k_, v_ in rec.get("kernels", {}).items() if "sparse_attn" in k_) or None
                traffic_note = f"profiles/{name}: " + rec.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command")
                break
        if traffic is None and why:
            traffic_note = "no matching PMC record: " + "; ".join(why[:4])
        gname = "sparse_attn_rows80_kernel" if d == 80 and esz == 2 else "sparse_attn_rows_kernel"
        if fused_ij:      # steps I + J in one launch: the kernel also expands the kept pixels and writes the CSR's columns
            gname += " (fused form: interpolation + attention, sea_sparse_attention_fused)"
        kname = {"tile": "sparse_attn_tile_kernel", "gather": gname, "auto": gname} if fused_ij else {"tile": "sparse_attn_tile_kernel", "gather": gname,
                 "auto": "attn_plan_kernel + " + gname + " + sparse_attn_tile_kernel (kernel choice on the device: the plan, the "
                         "running kernel and the idle kernel's exit all sit inside the timed events)"}
        kname = kname[path]
        # achieved = SURVEY 8d's algorithmic bytes (every gathered K / V row counted once per entry) / launch time.  K + V of
        # one head stay in the XCD's L2, so what binds is the L2 -> L1 request path, not HBM: `bound` says so, `frac` is
        # still against the 8 TB/s HBM line (the contract's roofline), `l2_gather_frac` against the 17.8 TB/s the
        # microarchitecture guide measures for L2-served row gathers, `frac_compulsory_hbm` = bytes that MUST cross HBM once
        return {"bound": "l2_gather", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                "compulsory_bytes": compulsory, "frac_compulsory_hbm": round(compulsory / t_attn_s / 1e9 / HBM_PEAK_GBS, 4),
                "l2_gather_peak": L2_GATHER_GBS, "l2_gather_frac": round(achieved / L2_GATHER_GBS, 4),
                # what the L2 actually served (PMC TCC_REQ x 128 B per launch, same record as `traffic`) against its streaming peak
                "l2_request_bytes": (l2_req * 128 if l2_req else None), "l2_peak": L2_PEAK_GBS,
                "frac_l2_peak": (round(l2_req * 128 / t_attn_s / 1e9 / L2_PEAK_GBS, 4) if l2_req else None),
                "timing": timing_note, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(t_attn_s * 1e3, 4), "nnz": Z}, Z

    def release(self):
        self.graph = self.rec = self.g_out = None
        self.attn_events.clear()
        for n_ in ("layer", "q", "k", "v", "mask"):
            setattr(self, n_, None)
        self.torch.cuda.empty_cache()


def short_leg(wname, NB, args, dev, ctx_dtype_name=None, dtype_name=None, override=None, sparse_kernel=None, steps=None,
              want_regions=False, layer_attrs=None):
    """A short run of another BASELINE shape (or of the headline with another context dtype / data dtype / predictor length):
    same protocol as the headline (prewarm, A/B of the attention path, HIP-graph replay, HIP events around the attention
    launch).  `sparse_kernel`: a fixed path instead of the A/B (the grid legs); `want_regions`: per-region times of a few
    eager steps after the timed ones."""
    dtype_name = dtype_name or args.dtype
    steps = steps or args.other_steps
    ctx_dtype_name = ctx_dtype_name or ("fp32" if args.context_dtype == "fp32" else dtype_name)
    lb = LayerBench(wname, NB, dtype_name, dev, ctx_dtype_name=ctx_dtype_name, override=override, layer_attrs=layer_attrs)
    try:
        for _ in range(3):
            lb.forward()
        path, ab = ("eager", {"graph": False}) if args.eager else lb.choose_path(sparse_kernel or args.sparse_kernel)
        if lb.graph is None:
            lb.layer.attention.sparse_kernel = "auto" if path in ("ab", "eager") else path
            path = lb.layer.attention.sparse_kernel
        lb.timed(max(2, args.warmup // 2))
        dt, am = lb.timed(steps)
        out, _ = lb.step()
        lb.torch.cuda.synchronize()
        roof, Z = lb.roofline(out, am / 1e3 if am else None, path, "HIP events around every launch inside the timed steps")
        w = lb.w
        res = {"workload": f"{wname} H={w['H']} d={w['d']} T={w['T']} k={w['k']} predictor_length={w['T_M']}, batch {NB}/GPU, "
                           f"{dtype_name}, context_layer {ctx_dtype_name}",
               "ms_per_step": round(dt / steps * 1e3, 4), "tokens_per_s": round(NB * w["T"] / (dt / steps), 1),
               "steps": steps, "graph": lb.graph is not None, "attention_path": path,
               "attention_path_ab": ab.get("candidates"), "roofline": roof}
        if want_regions:
            bench = lb.S.get_bench()
            bench.disabled, bench.synchronize = False, True
            bench.reset_measures()
            lb.layer.attention.sparse_kernel = path
            for _ in range(3):
                lb.forward()
            lb.torch.cuda.synchronize()
            res["regions_ms"] = {k_: round(v_ * 1e3, 4) for k_, v_ in sorted(bench.todict().items())}
            bench.disabled, bench.synchronize = True, False
            bench.reset_measures()
        return res
    finally:
        lb.release()


# the reference's own ablation grid: src/main/benchmark_opt_ablation.py:160-186 (one opt-125m layer, bsize 1, seq_len 2048,
# ks x ws) + the long-context point of src/main/exp_long_context.py:152 (T_M = 96, k = 128; opt-125m, 4096 tokens)
GRID_POINTS = [(2048, k_, w_) for k_ in (32, 64, 128) for w_ in (64, 128, 256, 384)] + [(4096, 128, 96)]


def grid_leg(args, dev):
    pts = {}
    for T_, k_, w_ in GRID_POINTS:
        try:
            r = short_leg("opt-125m", 1, args, dev, override=dict(T=T_, k=k_, T_M=w_), sparse_kernel="auto", steps=max(10, args.other_steps))
            pts[f"k:{k_},w:{w_},l:{T_}"] = {"ms_per_step": r["ms_per_step"], "attention_ms": (r["roofline"] or {}).get("avg_launch_ms"),
                                           "graph": r["graph"], "nnz": (r["roofline"] or {}).get("nnz")}
        except Exception as e:
            pts[f"k:{k_},w:{w_},l:{T_}"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return {"protocol": "src/main/benchmark_opt_ablation.py:160-186 (perlin, nbf 8, one opt-125m layer H=12 d=64, batch 1, T=2048; k x w) + "
                        "exp_long_context.py:152 (T_M=96, k=128, T=4096); latency of the SEA attention layer forward in ms, "
                        f"{args.dtype} data, context_layer fp32 (module defaults), HIP-graph replay + eager attention launch; every "
                        "point on the fused estimator kernels (tests/test_gpu_grid.py asserts it)",
            "points": pts}


def reference_fixture_check(dev):
    """The hot path (steps H..K) against the REFERENCE's own outputs, inside the bench process: `tests/golden/*.npz` hold seeded
    inputs and what the reference's Triton / torch operators returned for them (`tests/golden/make_golden.py`, run once in the
    build container; data only).  Selection + interpolation: CSR row pointers and column ids bit-exact; fused sparse attention
    (one launch instead of the reference's four operators): fp32 result within 2e-5 abs / 1e-4 rel.  The same comparisons as
    `tests/test_gpu_golden.py`, on the cases that fit a second."""
    import numpy as np
    import torch
    from sea_attention_amd.perlin_attention import ops
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    res = {"cases": [], "csr_bit_exact": True, "attention_max_abs_err": 0.0, "attention_within_tolerance": True}
    for case in ("large", "big", "ragged", "clamp"):
        f = os.path.join(gdir, case + ".npz")
        if not os.path.exists(f):
            continue
        g = np.load(f)
        N, H, T_DST, T_SRC, T_M, k, d, causal = [int(x) for x in g["meta"]]
        if not causal:
            continue
        keep = ops.keep_table_kernel_test(H, T_DST, T_M, k, device=dev)
        probs = torch.from_numpy(g["probs"]).to(dev)
        csr, _ = ops.topk_to_csr(probs, keep, k, target_width=T_SRC, is_causal=True)
        t = csr.to_sparse_csr()
        ok = (np.array_equal(t.crow_indices().cpu().numpy(), g["crow"]) and np.array_equal(t.col_indices().cpu().numpy(), g["col"]))
        res["csr_bit_exact"] &= bool(ok)
        q, kk, v = (torch.from_numpy(g[n_]).to(dev) for n_ in ("q", "k", "v"))
        out = ops.sparse_attention(q, kk, v, csr, row_scale=torch.from_numpy(g["scaler"]).to(dev).contiguous()).cpu().numpy()
        err = float(np.abs(out - g["sdbmm"]).max())
        res["attention_max_abs_err"] = max(res["attention_max_abs_err"], err)
        res["attention_within_tolerance"] &= bool(np.allclose(out, g["sdbmm"], atol=2e-5, rtol=1e-4))
        res["cases"].append(f"{case}: N{N} H{H} T{T_DST} T_M{T_M} k{k} d{d}, nnz {int(g['crow'][:, -1].sum())}")
    res["attention_max_abs_err"] = float(f"{res['attention_max_abs_err']:.3g}")
    res["status"] = "ok" if (res["cases"] and res["csr_bit_exact"] and res["attention_within_tolerance"]) else ("skipped: no fixtures" if not res["cases"] else "FAILED")
    res["note"] = ("tests/golden/*.npz = the reference's own operator outputs on seeded inputs (fp32 data): CSR indices bit-exact, "
                   "fused attention within 2e-5 abs / 1e-4 rel")
    return res


def decode_leg(layer, q, kk, v, mask, NB, T, positions):
    """Generation (the reference's loop: src/main/opt_generate.py:131): one position per step from a prefix, the
    step's launches replayed as one HIP graph (DecodeSession); every sequence of the batch advances together."""
    import copy
    import torch
    from sea_attention_amd.perlin_attention.decode import DecodeSession
    lc_ = copy.deepcopy(layer)
    lc_.pconfig = copy.copy(layer.pconfig); lc_.pconfig.use_cache = True
    lc_.attention.pconfig = lc_.pconfig
    nd = max(1, min(positions, T // 2))
    passes = 3 if 3 * nd + 8 <= T // 2 else 1               # the median pass is reported: one host hiccup (allocator, GC) inside a
    T0 = T - passes * nd - 8                                 # 32-position pass once read 2.2 ms per position instead of 0.075
    with torch.no_grad():
        pre = lc_(None, None, None, query_layer=q[:NB, :, :T0], key_layer=kk[:NB, :, :T0], value_layer=v[:NB, :, :T0],
                  attention_mask=mask[:NB, :, :T0, :T0].contiguous())
        sess = DecodeSession(lc_.attention, pre.state, kk[:NB, :, :T0], v[:NB, :, :T0], capacity=T, use_graph=True)
        for i in range(4):
            sess.step(q[:NB, :, T0 + i:T0 + i + 1], kk[:NB, :, T0 + i:T0 + i + 1], v[:NB, :, T0 + i:T0 + i + 1])
        per_pass = []
        for ps in range(passes):
            torch.cuda.synchronize(); t0_ = time.perf_counter()
            for i in range(4 + ps * nd, 4 + (ps + 1) * nd):
                sess.step(q[:NB, :, T0 + i:T0 + i + 1], kk[:NB, :, T0 + i:T0 + i + 1], v[:NB, :, T0 + i:T0 + i + 1])
            torch.cuda.synchronize()
            per_pass.append((time.perf_counter() - t0_) / nd)
        t_pos = sorted(per_pass)[len(per_pass) // 2]
        captures = getattr(sess, "captures", None)
    del sess, lc_, pre
    return {"ms_per_position": round(t_pos * 1e3, 4), "tokens_per_s": round(NB / t_pos, 1), "batch": NB, "prefix_tokens": T0,
            "positions_timed": nd, "passes_ms_per_position": [round(t * 1e3, 4) for t in per_pass], "graph_captures": captures}


def train_step_leg(wname, args, dev):
    """Forward + backward of the layer's sparse branch on ONE sequence of the headline shape: the HIP forward
    with saved per-entry probabilities and the HIP backward (dQ by rows, dK / dV through the transposed CSR), gradients
    to q, k, v.  The estimator runs without autograd (its map is an input of the branch, as in the reference's
    sparse-mode training, attention.py:1034-1042)."""
    import torch
    from sea_attention_amd.perlin_attention import ops
    lb = LayerBench(wname, 1, args.dtype, dev)
    try:
        out = lb.forward()
        csr = out.partial_attention_mask
        w = lb.w
        H, d, T = w["H"], w["d"], w["T"]
        q, k, v = (t_.detach().clone().requires_grad_(True) for t_ in (lb.q, lb.k, lb.v))
        rs = torch.sigmoid(torch.randn((1, H, T), device=dev)).requires_grad_(True)
        go = torch.randn((1, H, T, d), device=dev, dtype=torch.float32)

        def one():
            for t_ in (q, k, v, rs):
                t_.grad = None
            o = ops.sparse_attention_autograd(q, k, v, csr, row_scale=rs)
            o.backward(go)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_it = 10
        e0.record()
        for _ in range(n_it):
            one()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n_it
        fin = all(bool(torch.isfinite(t_.grad.float()).all().item()) for t_ in (q, k, v, rs))
        Z = int(csr.crow[:, -1].sum().item())
        return {"workload": f"{wname} sparse branch forward + backward, 1 sequence x {T} tokens, {args.dtype}, nnz {Z}",
                "ms_per_step": round(ms, 4), "tokens_per_s": round(T / (ms / 1e3), 1), "grads_finite": fin,
                "note": "ops.sparse_attention_autograd: sea_sparse_attention_ex (per-entry probabilities saved) + "
                        "sea_sparse_attention_bwd_gather (dK / dV gathered over the transposed pattern, no float atomics)"}
    finally:
        lb.release()


# ---------------------------------------------------------------------------------------------------------------------
def cpu_rehearsal(args, world, rank):
    """--rehearse on a box with NO GPU: the launcher, WORLD_SIZE check, process group (gloo), batch shards, the pipelined
    in-place all-gather of context shards, barrier + max-over-ranks timing and the single JSON line are the real ones;
    the compute step is a stand-in (each rank fills its shard with a rank/step pattern).  The numbers mean nothing."""
    import torch
    import torch.distributed as dist
    from sea_attention_amd import distributed as D
    w = WORKLOADS[args.workload]
    NB, T, C = args.batch, 64, 32
    chunks = max(1, args.gather_chunks)
    assert NB % chunks == 0, "--gather-chunks must divide the per-GPU batch"
    gather = (D.ChunkedContextGatherer((NB, T, C), NB * world, torch.float32, "cpu", chunks=chunks) if chunks > 1
              else D.ContextGatherer((NB, T, C), NB * world, torch.float32, "cpu"))
    ok = True

    def step(i):
        slot = gather.next_slot()
        if chunks > 1:
            for c in range(chunks):
                gather.local_chunk(slot, c).fill_(float(1000 * rank + i))
                gather.launch_chunk(slot, c)
            return slot, None
        gather.local[slot].fill_(float(1000 * rank + i))
        return slot, gather.launch(slot)
    for i in range(args.warmup):
        step(i)
    gather.finish()
    dist.barrier()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(i)
    gather.finish()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    full = last[1] if chunks == 1 else gather.gathered_items(last[0])
    for r in range(world):
        ok &= bool((full[r * NB:(r + 1) * NB] == float(1000 * r + args.steps - 1)).all())
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": METRIC, "value": round(NB * world * T / (elapsed / args.steps), 1), "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "rehearsal": "cpu-stand-in (no GPU on this box): launcher, process group (gloo), shards, pipelined all-gather and "
                         "timing protocol only -- the numbers mean nothing",
            "config": {"workload": f"REHEARSAL of {args.workload} (H={w['H']} d={w['d']} T={w['T']}): stand-in step",
                       "global_batch": NB * world, "seq_len": T, "parallelism": f"dp{world} (batch shards)"},
            "collective": {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "chunks_per_step": chunks},
            "output_check": {"status": "ok" if flag.item() == 1.0 else "FAILED", "gathered_shards_match_their_ranks": bool(flag.item() == 1.0)},
            "roofline": None, "cpu_baseline": None}))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if flag.item() == 1.0 else 4


def _stage(msg):
    """progress marker on stderr (a fault in a later stage is then attributed to it); synchronises first"""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
    print(f"[bench] {time.strftime('%H:%M:%S')} {msg}", file=sys.stderr, flush=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        print("[bench] --gpus must be >= 1", file=sys.stderr)
        return 2
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return self_launch(args, argv)                          # before anything here touches the GPU
    world = int(env_world or "1")
    if args.context_dtype == "auto":
        args.context_dtype = "fp32" if world == 1 else "same"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                                      # never print a line that claims GPUs that did not run
        print(f"[bench] rank {rank}: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to run",
              file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist
    have_gpu = torch.cuda.is_available()
    rccl_log = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            local_rank = 0
        if have_gpu:
            torch.cuda.set_device(local_rank)
        if not args.rehearse:
            # RCCL reports the algorithm / protocol / transports it picks into a per-process file; rank 0 parses its own
            # after the run (`collective.rccl`).  setdefault: a caller's own NCCL_DEBUG settings win
            rccl_log = os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/sea_bench_rccl_{os.getpid()}_%h_%p.log")
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING")
        dist.init_process_group("gloo" if args.rehearse else "nccl", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    if args.rehearse and not have_gpu:
        if world == 1:
            print("[bench] --rehearse without a GPU needs --gpus > 1 (it rehearses the multi-rank protocol)", file=sys.stderr)
            return 2
        return cpu_rehearsal(args, world, rank)
    assert have_gpu, "bench.py measures the HIP path; it needs the MI355X"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    import sea_attention_amd as S
    from sea_attention_amd import _lib, distributed as D
    from sea_attention_amd.perlin_attention import ops
    _lib.load(build_if_missing=True)

    w = dict(WORKLOADS[args.workload], **({"T": args.seq_len} if args.seq_len else {}))
    H, d, T, T_M, k = w["H"], w["d"], w["T"], w["T_M"], w["k"]
    NB = args.batch
    lb = LayerBench(args.workload, NB, args.dtype, dev, ctx_dtype_name="fp32" if args.context_dtype == "fp32" else args.dtype,
                    inspect_padding=args.inspect_padding, seed_offset=rank, override={"T": args.seq_len} if args.seq_len else None)
    dtype = lb.dtype
    layer, q, kk, v, mask = lb.layer, lb.q, lb.k, lb.v, lb.mask
    bench = S.get_bench()

    # N > 1: the all-gather of step i (RCCL, its own stream) overlaps the compute of step i+1 -- two slots of
    # (local shard, gathered output); every collective is awaited before the timed region ends (gather.finish()).
    chunks = max(1, args.gather_chunks) if world > 1 else 1
    if chunks > 1 and NB % chunks:
        print(f"[bench] --gather-chunks {chunks} does not divide the per-GPU batch {NB}", file=sys.stderr)
        return 2
    gather = None
    if world > 1:
        gather = (D.ChunkedContextGatherer((NB, T, H * d), NB * world, lb.ctx_dtype, dev, chunks=chunks) if chunks > 1
                  else D.ContextGatherer((NB, T, H * d), NB * world, lb.ctx_dtype, dev))

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def slot_view(slot):
        return gather.local[slot].view(NB, T, H, d).permute(0, 2, 1, 3)

    def step(do_gather=True):
        """One step of the job: the layer (graph replay + attention launch, or eager) and -- N > 1 -- the asynchronous
        all-gather of this step's context shard."""
        if world == 1:
            return lb.step()
        slot = gather.next_slot()
        if chunks > 1:
            n_c = NB // chunks
            views = [gather.local_chunk(slot, c).view(n_c, T, H, d).permute(0, 2, 1, 3) for c in range(chunks)]
            send = (lambda c: gather.launch_chunk(slot, c)) if do_gather else None
            if lb.graph is not None:                       # one attention launch + one collective per group of sequences
                out, _ = lb.step(chunk_views=views, after_chunk=send)
            else:
                out, ctx_ = lb.step()
                for c in range(chunks):
                    gather.local_chunk(slot, c).copy_(ctx_[c * n_c:(c + 1) * n_c])
                    if send is not None:
                        send(c)
            return out, (lambda: gather.gathered_items(slot))
        if lb.graph is not None:                           # the attention kernel writes straight into this step's gather slot
            out, _ = lb.step(out_view=slot_view(slot))
        else:
            out, ctx_ = lb.step()
            gather.local[slot].copy_(ctx_)
        full = gather.launch(slot) if do_gather else gather.out[slot]
        return out, full

    _stage("layer built; prewarm")
    for _ in range(args.prewarm):
        lb.forward()
    _stage("attention path A/B + graph capture")
    # kernel path of the attention launch + HIP-graph capture (both outside the timed region)
    if gather is not None:
        gather.finish()
    sync_all()
    if args.eager:
        path = "auto" if (args.sparse_kernel == "ab" and lb.tile_supported()) else ("gather" if args.sparse_kernel == "ab" else args.sparse_kernel)
        lb.layer.attention.sparse_kernel = path
        ab = {"requested": args.sparse_kernel, "graph": False}
    else:
        path, ab = lb.choose_path(args.sparse_kernel, with_process_group=world > 1)
        if world > 1:       # every rank runs the same kernel: rank 0's choice (the candidates differ by a few % between GPUs)
            names = ["gather", "auto", "tile"]
            c = torch.tensor([names.index(path)], device=dev)
            dist.broadcast(c, 0)
            if names[int(c.item())] != path:
                path = names[int(c.item())]
                lb.capture(path, with_process_group=True)
        if lb.graph is None:
            path = "auto" if lb.tile_supported() else "gather"
            lb.layer.attention.sparse_kernel = path
    _stage(f"path {path}; warmup + timed steps")
    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.finish()
    # per-kernel HIP events through the module's named regions (events only, no host sync inside the steps)
    bench.disabled, bench.synchronize = (lb.graph is not None), True   # regions cannot be timed inside a graph replay
    bench.reset_measures()
    lb.attn_events.clear()
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
    t_attn_graph = (lb.attn_ms() or 0) / 1e3 if lb.graph is not None else None

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    tokens_per_s = NB * world * T / (elapsed / args.steps)

    # ---- the same K steps again, `--repeats` times (outside the timed region that `value` comes from): K x 2 ms is thin against
    # box-to-box and run-to-run variance, so the line also says how stable the number is (min / median / max over all runs)
    repeats = None
    if args.repeats > 0:
        runs = [ms_per_step]
        for _ in range(args.repeats):
            sync_all()
            tr0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            if gather is not None:
                gather.finish()
            sync_all()
            tr = torch.tensor([time.perf_counter() - tr0], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            runs.append(float(tr.item()) / args.steps * 1e3)
        srt = sorted(runs)
        repeats = {"runs": len(runs), "steps_each": args.steps, "ms_per_step_min": round(srt[0], 4),
                   "ms_per_step_median": round(srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2]), 4),
                   "ms_per_step_max": round(srt[-1], 4), "ms_per_step_all": [round(r_, 4) for r_ in runs],
                   "note": "run 0 is the timed region `value` / `ms_per_step` report; the others repeat it back to back (same barrier + "
                           "synchronize bracket, max over ranks)"}

    # ---- N > 1: the collective by itself and the compute by itself (outside the timed region) -------------------------
    collective = None
    if world > 1:
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(do_gather=False)
        torch.cuda.synchronize()
        t_comp = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(t_comp, op=dist.ReduceOp.MAX)
        sync_all()
        t2 = time.perf_counter()
        for _ in range(args.steps):                        # back-to-back all-gathers of one slot, each awaited
            if chunks > 1:
                for c in range(chunks):
                    gather.launch_chunk(0, c)
            else:
                gather.launch(0)
            gather.finish()
        torch.cuda.synchronize()
        t_ag = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
        dist.all_reduce(t_ag, op=dist.ReduceOp.MAX)
        rccl_info = None
        if rank == 0 and rccl_log is not None:             # what RCCL says it chose (algorithm / protocol / transports)
            try:
                import glob
                txt = "".join(open(f_, errors="replace").read() for f_ in sorted(glob.glob(rccl_log.replace("%h", "*").replace("%p", "*"))))
                rccl_info = D.parse_nccl_debug(txt)
            except Exception as e_:
                rccl_info = {"error": f"{type(e_).__name__}: {e_}"[:200]}
        comp_ms = float(t_comp.item()) / args.steps * 1e3
        ag_ms = float(t_ag.item()) / args.steps * 1e3
        shard_bytes = NB * T * H * d * torch.tensor([], dtype=lb.ctx_dtype).element_size()
        collective = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                      "compute_only_ms_per_step": round(comp_ms, 4),
                      "allgather_alone_ms": round(ag_ms, 4),
                      "allgather_exposed_ms_per_step": round(max(0.0, ms_per_step - comp_ms), 4),
                      "chunks_per_step": chunks, "rccl": rccl_info,
                      "shard_bytes": shard_bytes, "received_bytes_per_rank": shard_bytes * (world - 1),
                      "allgather_alone_GBs_per_rank": round(shard_bytes * (world - 1) / (ag_ms / 1e3) / 1e9, 1),
                      "note": "compute_only = the same steps without launching the collective (the N=1-equivalent per-GPU step, "
                              "max over ranks); allgather_alone = all_gather_into_tensor of one step's shards, awaited, nothing "
                              "else running; exposed = ms_per_step - compute_only (the pipelined gather hides the rest)"}

    if lb.graph is not None:
        bench.disabled = False                            # informational per-region times: a few eager steps
        lb.layer.attention.sparse_kernel = path
        for _ in range(5):
            lb.forward()
        torch.cuda.synchronize()
        regions = bench.todict()
    bench.disabled, bench.synchronize = True, False
    bench.reset_measures()

    _stage("timed region done; output check")
    # ---- output self-check (outside the timed region): the batch the bench times is also a batch that is RIGHT -------
    # item n of the batched output == the same item run ALONE through the whole layer, bit for bit (every kernel is
    # deterministic and treats batch items independently; the Performer is told to take the one-pass kernel the batch
    # takes instead of cutting the lone sequence into segments), and its CSR is rebuilt from its map by the unfused
    # top-k path as a second opinion
    output_check = None
    if not args.no_output_check:
        with torch.no_grad():
            # N > 1 in graph mode: the attention launch wrote this rank's shard straight into the gathered buffer
            if callable(ctx):                              # chunked gather: the rank-major copy of the last step's slot
                ctx = ctx()
            ctx_b = ctx[rank * NB:(rank + 1) * NB] if world > 1 else out.context_layer
            probs_b = out.estimated_attention_probs_m
            keep_t = ops.keep_table_causal(H, T, T_M, k, device=dev)
            zc = ops.z_capacity(keep_t.cpu(), H, T, T, T_M, k, True)
            bits_ok, worst = True, 0.0
            # the lone item takes the Performer path the BATCH took (one pass for a full batch; for a one-sequence workload
            # the library's sequence-parallel plan of that very shape)
            layer.attention.performer_segments = ops.performer_plan(NB, H, T, d, layer.attention.performer_nb_features, dtype)[0]
            layer.attention.sparse_kernel = path
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
                # 'auto' picks the kernel from the whole launch's plan (a lone item can fall on the other side of the cut):
                # equal to rounding there, bitwise for a fixed kernel path
                close = worst < 2e-3 if path == "auto" else bool(torch.equal(alone.context_layer, ctx_b[n_:n_ + 1]))
                bits_ok &= same_map and close
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
                            "items_alone_equal_to_batched_rows": bits_ok,
                            "comparison": "bitwise" if path != "auto" else "2e-3 rel-norm (the kernel choice may differ between a lone item and the batch)",
                            "finite": finite, "items_checked": sorted({0, NB - 1}), "layer_item_alone_rel_diff": round(worst, 8)}
            if gathered_ok is not None:
                output_check["gathered_shards_match_their_ranks"] = gathered_ok
            if rank == 0:
                try:
                    output_check["reference_fixtures"] = reference_fixture_check(dev)
                    if output_check["reference_fixtures"]["status"] == "FAILED":
                        output_check["status"] = "FAILED"
                except Exception as e:                       # (a missing fixture directory must not take the line down)
                    output_check["reference_fixtures"] = {"status": f"error: {type(e).__name__}: {e}"[:200]}

    # ---- roofline of the dominant HIP kernel (fused sparse attention) ---------------------------------
    t_attn = t_attn_graph if t_attn_graph else regions.get('attention.sparse.fused')
    roof, Z = lb.roofline(out, t_attn, path,
                          "HIP events around every launch inside the timed steps" if lb.graph is not None
                          else "HIP events of the module's 'attention.sparse.fused' region inside the timed steps")
    esz = torch.tensor([], dtype=dtype).element_size()

    _stage("kernel-level leg")
    # ---- kernel-level path only (H..K on HIP, probs given) -- what the CPU baseline below also runs -------
    kernel_path = None
    if args.kernel_iters > 0:
        kpath = path
        probs = torch.softmax(torch.randn((NB, H, T, T_M), device=dev), -1).to(dtype)
        keep = ops.keep_table_causal(H, T, T_M, k, device=dev)
        z_cap = ops.z_capacity(keep.cpu(), H, T, T, T_M, k, True)
        rs = torch.sigmoid(torch.randn((NB, H, T), device=dev))
        mx = torch.sigmoid(torch.randn((NB, H, T), device=dev))
        avg = (v.float().cumsum(-2) / torch.arange(1, T + 1, device=dev).view(1, 1, -1, 1)).to(dtype)
        ctx2 = torch.empty((NB, T, H * d), dtype=dtype, device=dev)

        def kstep():
            c, _ = ops.topk_to_csr(probs, keep, k, target_width=T, z_cap=z_cap)
            ops.sparse_attention(q, kk, v, c, row_scale=rs, avg=avg, mix=mx, out=ctx2.view(NB, T, H, d).permute(0, 2, 1, 3), path=kpath)
            return c
        for _ in range(3):
            kstep()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        tk = ta = 0.0
        for _ in range(args.kernel_iters):
            e0.record(); c, _ = ops.topk_to_csr(probs, keep, k, target_width=T, z_cap=z_cap); e1.record()
            ops.sparse_attention(q, kk, v, c, row_scale=rs, avg=avg, mix=mx, out=ctx2.view(NB, T, H, d).permute(0, 2, 1, 3), path=kpath)
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
            del cs
        del probs, avg, ctx2, rs, mx

    _stage("generation leg / CPU baseline")
    # ---- generation leg: one position per step from a (T - 64)-token prefix, the step replayed as a HIP graph
    decode = None
    if args.decode_steps > 0 and world == 1 and dtype != torch.float32:
        decode = {"note": "DecodeSession: fixed-capacity caches, position in device memory, the step's launches replayed as one HIP "
                          "graph; all sequences of a batch advance together (reference loop: src/main/opt_generate.py:131)"}
        for nb_ in sorted({1, NB}):
            try:
                decode[f"batch_{nb_}"] = decode_leg(layer, q, kk, v, mask, nb_, T, args.decode_steps)
            except Exception as e:                                   # an extra leg never takes the headline line down
                decode[f"batch_{nb_}"] = {"error": f"{type(e).__name__}: {e}"[:200]}

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

    if cpu is not None:
        # which GPU number each CPU number is the twin of (the headline `value` is the WHOLE layer, A..L)
        fl = cpu.get("full_layer") or {}
        cpu["gpu_over_cpu"] = {
            "whole_layer": (round(tokens_per_s / fl["value"], 1) if fl.get("value") else None),
            "kernel_level": (round(kernel_path["value"] / cpu["value"], 1) if kernel_path and cpu.get("value") else None),
            "note": "whole_layer = `value` (steps A..L on the GPU) / cpu_baseline.full_layer (the same layer in the dense torch mode on "
                    "the host cores); kernel_level = kernel_path.value (steps H..K on the GPU, probs given) / cpu_baseline.value (the "
                    "oracle's dense port of the same steps).  `value` / cpu_baseline.value would compare a whole layer with a part of one"}
    graph_on = lb.graph is not None
    capture_error = lb.capture_error
    lazy_flags = (bool(getattr(lb.layer.attention, "lazy_attention_probs", False)), bool(getattr(lb.layer.attention, "lazy_csr_columns", False)))
    headline_workload = (f"{args.workload} SEA attention layer forward (sparse mode, steps A-L), "
                         f"H={H} d={d} T={T} k={k} predictor_length={T_M} nbf={w['nbf']}, "
                         f"batch {NB} sequences/GPU, random-init weights seed 42, "
                         + (f"context_layer fp32 (the module's and the reference's default, flat_csr_sdbmm.py:347; the {args.dtype} "
                            f"twin: context_{args.dtype}), " if args.context_dtype == "fp32" else
                            f"context_layer {args.dtype} (reference default: fp32"
                            + ("; N > 1: the shard every rank all-gathers is 16-bit, half the bytes over xGMI" if world > 1 else "") + "), ")
                         + ("unpadded batch declared to the module (assume_not_padded: no mask inspection sync)"
                            if not args.inspect_padding else "module inspects the mask for padding (one host sync)")
                         + f", sparse kernel path {path}"
                         + (f", module defaults lazy_attention_probs={lazy_flags[0]} / lazy_csr_columns={lazy_flags[1]} (the (N,H,T,T_M) "
                            "probability map and the CSR's column array -- outputs the reference materialises every step, "
                            "attention.py:1343, causal_resize_m_to_t.py:757-762 -- are computed on first read, not by the step: the "
                            "`eager_outputs` leg times the step that writes both)" if any(lazy_flags) else
                            ", eager outputs (probability map and CSR columns written every step, as the reference does)")
                         + (f", + {'gloo (rehearsal)' if args.rehearse else 'RCCL'} all-gather of context shards" if world > 1 else "")
                         + (", layer replayed as a HIP graph + eager fused-attention launch" if graph_on else ", eager launches"))

    # ---- the other BASELINE shapes + the reference-default (fp32 context) twin of the headline: short legs, rank 0's GPU ----
    other, ctx_twin, train, fp32_leg, grid, long_ctx, eager_out = None, None, None, None, None, None, None
    del out, ctx, q, kk, v, mask, layer
    lb.release()
    if gather is not None:
        del gather
    torch.cuda.empty_cache()
    _stage("short legs")
    twin_name = f"context_{args.dtype}" if args.context_dtype == "fp32" else "context_fp32"
    if world == 1 and not args.no_other_workloads:
        other = {}
        try:   # the headline with the OTHER context dtype: the tuned 16-bit output (default run) or the reference's fp32
            ctx_twin = short_leg(args.workload, NB, args, dev, ctx_dtype_name=args.dtype if args.context_dtype == "fp32" else "fp32")
            ctx_twin["tokens_per_s_note"] = ("the headline layer writing context_layer in the data dtype (a tuned configuration: the module "
                                             "default is fp32)" if args.context_dtype == "fp32" else
                                             "the headline layer writing context_layer in fp32, the reference's default (flat_csr_sdbmm.py:347)")
        except Exception as e:
            ctx_twin = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_eager_outputs:
            _stage("eager-outputs leg")
            try:   # the headline with the reference's eager outputs: the map and the column array written by every step
                eager_out = short_leg(args.workload, NB, args, dev, ctx_dtype_name="fp32" if args.context_dtype == "fp32" else args.dtype,
                                      layer_attrs=dict(lazy_attention_probs=False, lazy_csr_columns=False))
                eager_out["note"] = ("lazy_attention_probs=False, lazy_csr_columns=False: estimated_attention_probs (N,H,T,T_M) and the flat "
                                     "CSR's col_indices are materialised inside every step, as the reference's benchmarking branch does "
                                     "(attention.py:1343, causal_resize_m_to_t.py:669,757-762); everything else as the headline")
                eager_out["ratio_to_headline"] = round(eager_out["ms_per_step"] / ms_per_step, 4)
            except Exception as e:
                eager_out = {"error": f"{type(e).__name__}: {e}"[:300]}
        for wn, nb_ in OTHER_LEGS:
            if wn == args.workload and nb_ == NB:
                continue
            _stage(f"leg {wn} x{nb_}")
            try:
                other[f"{wn} x{nb_}"] = short_leg(wn, nb_, args, dev)
            except Exception as e:                               # a leg never takes the headline down; its failure is visible
                other[f"{wn} x{nb_}"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_fp32_leg and args.dtype != "fp32":
            # BASELINE config 2 lists fp32 and the reference's measurement protocol is fp32 (benchmark_bert.py:196-239): the same
            # layer on fp32 DATA.  Steps H..L run on the HIP kernels; the estimator runs the fp32-MFMA Performer, library GEMMs
            # for the three Linears, the HIP LayerNorm / tail kernels and the framework's convolutions (no C8 form for fp32)
            _stage("fp32-data leg (config 2)")
            try:
                fp32_leg = short_leg("opt-125m", 8, args, dev, dtype_name="fp32", want_regions=True)
                ref16 = (other.get("opt-125m x8") or {}).get("ms_per_step")
                if ref16:
                    fp32_leg["ratio_to_16bit_data"] = round(fp32_leg["ms_per_step"] / ref16, 3)
            except Exception as e:
                fp32_leg = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_long_context and args.dtype != "fp32":
            # context extension (perlin_trainer.py:533-566): one 32768-token sequence, T / T_M = 128 > k -- every pixel of a late
            # row is thinned to max_k entries by the reference's fp32 stepping (causal_resize_m_to_t.py:565-569,657-659)
            _stage("long-context leg")
            try:
                long_ctx = short_leg("opt-125m", 1, args, dev, override=dict(T=32768))
            except Exception as e:
                long_ctx = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_grid and args.dtype != "fp32":
            _stage("reference grid leg")
            try:
                grid = grid_leg(args, dev)
            except Exception as e:
                grid = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_train_step and args.dtype != "fp32":
            _stage("train-step leg")
            try:
                train = train_step_leg(args.workload, args, dev)
            except Exception as e:
                train = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        line = {
            "metric": METRIC,
            "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": headline_workload, "global_batch": NB * world, "seq_len": T,
                       "parallelism": f"dp{world} (batch shards)"},
            "graph": graph_on, "graph_capture_error": capture_error,
            "roofline": roof, "cpu_baseline": cpu, "output_check": output_check, "attention_path_ab": ab,
            "repeats": repeats, "eager_outputs": eager_out,
            "collective": collective, twin_name: ctx_twin, "other_workloads": other, "fp32_data": fp32_leg,
            "reference_grid": grid, "long_context": long_ctx, "train_step": train,
            "kernel_path": kernel_path, "decode": decode,
            "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 3),
            "regions_ms": {k_: round(v_ * 1e3, 4) for k_, v_ in sorted(regions.items())},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if output_check is not None and output_check["status"] != "ok":
        return 5
    return 0


if __name__ == "__main__":
    sys.exit(main())
