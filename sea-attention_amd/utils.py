"""Observability hooks on the hot path: named timing regions and intermediate-buffer capture.

Same interface as the reference's `src/utils/__init__.py:384-537` (`get_bench().region(name)`,
`register_temp_buffer`, `format_tracetree`, ...): region names and buffer names inside the attention
module are a de-facto API for its benchmarks (`src/main/benchmark_bert.py:221-224`) and parity tests
(`src/main/tests/test_perlin_opt_consist.py:198-232`), so they are reproduced verbatim.
"""
import os
import random
import time

import numpy as np
import torch


def seed(seed=42):
    """src/utils/__init__.py:32-40"""
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def batch_to(batch, device):
    """src/utils/__init__.py:87-103: move a tensor / list / tuple / dict of tensors."""
    if isinstance(batch, torch.Tensor):
        return batch.to(device)
    if isinstance(batch, (list, tuple)):
        return type(batch)(batch_to(b, device) for b in batch)
    if isinstance(batch, dict):
        return {k: batch_to(v, device) for k, v in batch.items()}
    return batch


class _Region:
    def __init__(self, bench, name):
        self.bench, self.name = bench, name
        self.parent, self.children = None, []

    def __enter__(self):
        b = self.bench
        if b.disabled:
            return self
        self.t0 = time.time()
        if b.synchronize and torch.cuda.is_available():
            self.ev0 = torch.cuda.Event(enable_timing=True)
            self.ev0.record()
        if b.tracking_callstack:
            if b.current_region_context is not None:
                self.parent = b.current_region_context
                self.parent.children.append(self)
            b.current_region_context = self
        return self

    def __exit__(self, *exc):
        b = self.bench
        if b.disabled:
            return False
        if b.synchronize and torch.cuda.is_available():
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            ev0 = self.ev0

            def measure():
                torch.cuda.synchronize()
                return ev0.elapsed_time(ev1) / 1000
            b.add_data(self.name, measure)
        else:
            b.add_data(self.name, time.time() - self.t0)
        if b.tracking_callstack:
            if self.parent is None:
                b.tracking_callstack = False
                b.current_region_context = None
                b.traced_callstack = self
            else:
                b.current_region_context = self.parent
        return False


class _MemRegion:
    def __init__(self, bench, name):
        self.bench, self.name = bench, name

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class Benchmark:
    def __init__(self):
        self.synchronize = False
        self.disabled = True
        self.activate_temp_buffers = False
        self.buffers = {}
        self.data = {}
        self.tracking_callstack = True
        self.current_region_context = None
        self.traced_callstack = None

    def add_data(self, name, t):
        count, total = self.data.get(name, (0, 0))
        if callable(t):
            total = (total if isinstance(total, list) else []) + [t]
        else:
            total = total + t
        self.data[name] = (count + 1, total)

    def reset_trace(self):
        self.tracking_callstack = True
        self.current_region_context = None
        self.traced_callstack = None

    def reset_measures(self):
        self.data = {}

    def region(self, name):
        return _Region(self, name)

    def mem_region(self, name):
        return _MemRegion(self, name)

    def todict(self):
        out = {}
        for key, (c, s) in self.data.items():
            if isinstance(s, list):
                s = sum(f() for f in s)
            out[key] = s / (c + 1e-10)
        return out

    def register_temp_buffer(self, name, v, lazy=None):
        if not self.activate_temp_buffers:
            return
        if v is None and lazy is not None:
            v = lazy()
        self.buffers.setdefault(name, []).append(v)

    def get_temp_buffer(self, name, index=-1):
        return self.buffers[name][index]

    def reset_temp_buffers(self):
        self.buffers = {}

    def format_tracetree(self):
        data = self.todict()
        root = self.traced_callstack
        if root is None:
            return ""
        total = data[root.name]

        def fmt(item, depth=0):
            pre = "" if depth == 0 else "  " * (depth - 1) + "╰─"
            lines = [f"{pre}> {item.name} ({data[item.name] * 1000:.2f} ms, {data[item.name] / total * 100:.2f}%)"]
            lines += [fmt(c, depth + 1) for c in item.children]
            return "\n".join(lines)
        return fmt(root)


BENCHMARK = Benchmark()


def get_bench() -> Benchmark:
    return BENCHMARK
