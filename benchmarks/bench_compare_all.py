#!/usr/bin/env python3
"""Sweep FA1 / FA2 / FA3 (HIP backend) over direction x N x d x B x H x causal x dtype — counterpart of
/root/reference/benchmarks/bench_compare_all.py:105-193 (and bench_fa{1,2,3}.py), same flags, same records.

    python benchmarks/bench_compare_all.py --seqlen 4096 --head-dim 128 --batch-size 8 --num-heads 32 --dtypes bf16

"backward" = clones of q, k, v with requires_grad + forward + out.sum().backward(), timed together
(bench_fa3.py:131-154).  fp32 tensors (and head dims that are not a multiple of 8) run on the exact-f32 kernels.
"""
import argparse
import sys

import torch
from bench_utils import (DTYPES, BenchmarkRecord, add_common_args, algorithmic_tflops, benchmark_fn, compute_tflops,
                         format_table, has_hip_extension, is_oom_error, iter_causal_flags, make_qkv, write_results)


def build_parser(default_tag="compare_all"):
    """The reference's flags, one for one (bench_compare_all.py:70-91, bench_fa3.py:50-70): the figure / table flags
    (--config-label, --plot-dtype, --no-plot, --caption) are accepted so that a reference command line runs unchanged;
    --config-label lands in the records' `config` field as it does there, the other three only matter to the plotting
    this repo does not rebuild (cosmetic).  --algos / --no-save are this repo's additions."""
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    add_common_args(ap)
    ap.add_argument("--fp8", action="store_true", help="Include fp8 path for FA3")
    ap.add_argument("--directions", nargs="+", default=["forward"], choices=["forward", "backward"],
                    help="Benchmark forward, backward, or both (default: forward, as the reference)")
    ap.add_argument("--tag", type=str, default=default_tag, help="Base name for result files")
    ap.add_argument("--config-label", type=str, default=None, help="Optional config name for tables (the records' config field)")
    ap.add_argument("--plot-dtype", type=str, default=None, help="accepted for compatibility: no figures are made here")
    ap.add_argument("--no-plot", action="store_true", help="accepted for compatibility: no figures are made here")
    ap.add_argument("--caption", type=str, default="Figure 6: Attention backward speed (FP16/BF16) on H100 GPU.",
                    help="accepted for compatibility: no figures are made here")
    ap.add_argument("--algos", nargs="+", default=["fa1", "fa2", "fa3"], choices=["fa1", "fa2", "fa3"])
    ap.add_argument("--no-save", action="store_true")
    return ap


def main(argv=None, default_tag="compare_all"):
    args = build_parser(default_tag).parse_args(argv)
    # The reference runs its torch backend on --device cpu; this build has no CPU compute path: every sweep
    # point is then recorded as an error, the table is still printed, and the exit code says so.
    no_backend = None
    if args.device != "cuda" or not has_hip_extension():
        no_backend = "the MI355X build has no CPU backend: run on a GPU box with the library built (make -C flashattention-pytorch_amd/csrc)"
    ops = {}
    if no_backend is None:
        from fa1.op import fa1_attention
        from fa2.op import fa2_attention
        from fa3.op import fa3_attention

        ops = {"fa1": fa1_attention, "fa2": fa2_attention, "fa3": fa3_attention}
    records = []
    for direction in args.directions:
        for n in args.seqlen:
            for d in args.head_dim:
                for b in args.batch_size:
                    for h in args.num_heads:
                        for causal in iter_causal_flags(args):
                            for dt in args.dtypes:
                                for algo in args.algos:
                                    for fp8 in ([False, True] if (algo == "fa3" and args.fp8) else [False]):
                                        if no_backend is not None:
                                            records.append(BenchmarkRecord(
                                                method=algo.upper(), algo=algo, backend="cuda", direction=direction, dtype=dt, causal=causal,
                                                seqlen=n, head_dim=d, batch_size=b, num_heads=h, mean_ms=None, std_ms=None, tflops=None,
                                                peak_mem_mb=None, status="error", fp8=fp8 if algo == "fa3" else None,
                                                config=args.config_label, error=no_backend))
                                        else:
                                            records.append(run_one(ops[algo], algo, direction, n, d, b, h, causal, dt, fp8, args))
    headers = ["method", "backend", "direction", "dtype", "shape", "mask", "mean ms", "std ms", "TFLOP/s (ref conv.)", "mem MB", "status"]
    print(format_table(headers, [r.to_row() for r in records]))
    if not args.no_save:
        print(write_results(args.tag, records))
    if no_backend is not None:
        sys.exit(no_backend)
    return records


def run_one(op, algo, direction, n, d, b, h, causal, dt, fp8, args):
    rec = dict(method=algo.upper(), algo=algo, backend="cuda", direction=direction, dtype=dt, causal=causal, seqlen=n,
               head_dim=d, batch_size=b, num_heads=h, fp8=fp8 if algo == "fa3" else None)
    try:
        q, k, v = make_qkv(b, h, n, d, "cuda", DTYPES[dt])
        kw = dict(causal=causal, softmax_scale=d ** -0.5, backend="cuda")
        if algo == "fa3":
            kw["fp8"] = fp8

        if direction == "forward":
            def call():
                with torch.no_grad():
                    return op(q, k, v, **kw)[0]
        else:
            def call():
                qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
                out = op(qq, kk, vv, **kw)[0]
                out.sum().backward()
                return out
        mean, std, mem = benchmark_fn(call, "cuda", args.warmup, args.iters)
        return BenchmarkRecord(mean_ms=mean, std_ms=std, tflops=compute_tflops(b, h, n, d, mean, direction), peak_mem_mb=mem,
                               status="ok", config=args.config_label if args.config_label is not None else
                               f"algorithmic {algorithmic_tflops(b, h, n, d, mean, direction, causal):.1f} TFLOP/s", **rec)
    except Exception as exc:  # noqa: BLE001 - a sweep records the failure and moves on (bench_fa2.py:136-139)
        torch.cuda.empty_cache()
        return BenchmarkRecord(mean_ms=None, std_ms=None, tflops=None, peak_mem_mb=None,
                               status="oom" if is_oom_error(exc) else "error", config=args.config_label, error=str(exc)[:200], **rec)


if __name__ == "__main__":
    main()
