"""Timing harness with the reference's record schema and CLI.

Counterpart of /root/reference/benchmarks/bench_utils.py: same synthetic inputs (`make_qkv`: seeded generator,
order q, k, v — :83-97), same timing protocol (`benchmark_fn`: warm-up, per-iteration perf_counter around the call
plus a device synchronise, mean / population stdev in ms, peak-memory delta — :100-146), same FLOP convention
(`attention_flops`: forward 4*B*H*N^2*d, "backward" = forward+backward timed together 8*B*H*N^2*d, no causal
discount — :210-224), same `BenchmarkRecord` fields and JSON/CSV writer (:161-207, :287-325) and the same common
flags (`add_common_args`, :247-263).  Nothing is created at import time (the reference mkdirs on import).
"""
from __future__ import annotations

import argparse
import csv
import json
import math
import os
import statistics
import sys
import time
from dataclasses import asdict, dataclass
from typing import Callable, List, Optional, Sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flashattention-pytorch_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import torch  # noqa: E402

RESULTS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "results")
FIELDS = ["method", "algo", "backend", "direction", "dtype", "causal", "seqlen", "head_dim", "batch_size", "num_heads",
          "mean_ms", "std_ms", "tflops", "peak_mem_mb", "status", "fp8", "config", "error"]
DTYPES = {"fp16": torch.float16, "float16": torch.float16, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
          "fp32": torch.float32, "float32": torch.float32}


@dataclass
class BenchmarkRecord:
    method: str
    algo: str
    backend: str
    direction: str
    dtype: str
    causal: bool
    seqlen: int
    head_dim: int
    batch_size: int
    num_heads: int
    mean_ms: Optional[float]
    std_ms: Optional[float]
    tflops: Optional[float]
    peak_mem_mb: Optional[float]
    status: str
    fp8: Optional[bool] = None
    config: Optional[str] = None
    error: Optional[str] = None

    def to_dict(self):
        return asdict(self)

    def to_row(self) -> List[str]:
        f = lambda x, p: "-" if x is None else f"{x:.{p}f}"  # noqa: E731
        return [self.method, self.backend, self.direction, self.dtype,
                f"B{self.batch_size} H{self.num_heads} N{self.seqlen} D{self.head_dim}",
                "causal" if self.causal else "non-causal", f(self.mean_ms, 2), f(self.std_ms, 2), f(self.tflops, 2),
                f(self.peak_mem_mb, 1), self.status if not self.error else f"{self.status} ({self.error})"]


def has_hip_extension() -> bool:
    if not torch.cuda.is_available():
        return False
    try:
        from fa2.cuda.impl import _load_ext

        _load_ext()
        return True
    except Exception:
        return False


def make_qkv(batch, heads, seqlen, head_dim, device, dtype):
    g = torch.Generator(device=device)
    g.manual_seed(0)
    shape = (batch, heads, seqlen, head_dim)
    return tuple(torch.randn(shape, device=device, dtype=dtype, generator=g) for _ in range(3))


def benchmark_fn(fn: Callable[[], object], device: str, warmup: int, iters: int):
    """(mean_ms, std_ms, peak_mem_mb) of fn(); synchronises the device inside every timed iteration."""
    cuda = device == "cuda"
    for _ in range(warmup):
        fn()
    if cuda:
        torch.cuda.synchronize()
    times, peak = [], None
    for _ in range(iters):
        if cuda:
            torch.cuda.reset_peak_memory_stats()
            start_mem = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        fn()
        if cuda:
            torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        if cuda:
            used = (torch.cuda.max_memory_allocated() - start_mem) / (1024 * 1024)
            peak = used if peak is None else max(peak, used)
    return statistics.mean(times), (statistics.pstdev(times) if len(times) > 1 else 0.0), peak


def attention_flops(batch, heads, seqlen, head_dim, direction="forward"):
    return (4.0 if direction == "forward" else 8.0) * batch * heads * seqlen * seqlen * head_dim


def compute_tflops(batch, heads, seqlen, head_dim, mean_ms, direction):
    if mean_ms is None or math.isnan(mean_ms):
        return None
    return attention_flops(batch, heads, seqlen, head_dim, direction) / (mean_ms / 1e3) / 1e12


def algorithmic_tflops(batch, heads, seqlen, head_dim, mean_ms, direction, causal):
    """SURVEY §8d convention: forward 4, forward+backward 14 (x N^2 d), causal x (N+1)/2N."""
    f = (4.0 if direction == "forward" else 14.0) * batch * heads * seqlen * seqlen * head_dim
    if causal:
        f *= (seqlen + 1) / (2.0 * seqlen)
    return f / (mean_ms / 1e3) / 1e12


def is_oom_error(exc: Exception) -> bool:
    return "out of memory" in str(exc).lower() or isinstance(exc, torch.cuda.OutOfMemoryError)


def add_common_args(p: argparse.ArgumentParser) -> None:
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu", choices=["cpu", "cuda"])
    p.add_argument("--seqlen", type=int, nargs="+", default=[512, 1024, 2048, 4096, 8192, 16384])
    p.add_argument("--head-dim", type=int, nargs="+", default=[64, 128, 256])
    p.add_argument("--batch-size", type=int, nargs="+", default=[1, 2])
    p.add_argument("--num-heads", type=int, nargs="+", default=[4])
    p.add_argument("--causal", action="store_true", help="run only causal (default: both)")
    p.add_argument("--non-causal-only", action="store_true")
    p.add_argument("--dtypes", type=str, nargs="+", default=["fp16", "bf16"])
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--iters", type=int, default=20)


def iter_causal_flags(args):
    if args.causal:
        return [True]
    if args.non_causal_only:
        return [False]
    return [False, True]


def format_table(headers: Sequence[str], rows: Sequence[Sequence[str]]) -> str:
    w = [max(len(h), *(len(r[i]) for r in rows)) if rows else len(h) for i, h in enumerate(headers)]
    line = lambda r: " | ".join(c.ljust(w[i]) for i, c in enumerate(r))  # noqa: E731
    return "\n".join([line(headers), "-+-".join("-" * x for x in w)] + [line(r) for r in rows])


def write_results(name: str, records: Sequence[BenchmarkRecord], out_dir: str = RESULTS_DIR):
    os.makedirs(out_dir, exist_ok=True)
    stem = os.path.join(out_dir, f"{name.replace(' ', '_')}_{time.strftime('%Y%m%d-%H%M%S')}")
    with open(stem + ".json", "w", encoding="utf-8") as f:
        json.dump([r.to_dict() for r in records], f, indent=2)
    with open(stem + ".csv", "w", newline="", encoding="utf-8") as f:
        wr = csv.DictWriter(f, fieldnames=FIELDS)
        wr.writeheader()
        for r in records:
            wr.writerow(r.to_dict())
    return {"json": stem + ".json", "csv": stem + ".csv"}


def load_results(paths: Sequence[str]) -> List[BenchmarkRecord]:
    out = []
    for p in paths:
        with open(p, encoding="utf-8") as f:
            out += [BenchmarkRecord(**d) for d in json.load(f)]
    return out
