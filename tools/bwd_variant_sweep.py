"""Backward variants by launch shape, interleaved in one process: default rule, stream dK/dV kernel forced (dkdv = 5), 8-wave
dK/dV kernel forced (dkdv = 8), dS hand-over forced (dq = 6), recomputing pass forced (dq = 5).  Backward ms (median of the
rounds; a pair of events around `--iters` calls), head_dim 128, bf16.

    python tools/bwd_variant_sweep.py [--causal] [--rounds 5] [--shapes 512x512 256x1024 ...]
"""
import argparse
import statistics
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

VARIANTS = [("default", {}), ("stream dK/dV", {"dkdv": 5}), ("8-wave dK/dV", {"dkdv": 8}), ("hand-over", {"dq": 6}), ("hand-over, 256-row dQ tiles", {"dq": 6, "dq_w4": 3}), ("recompute", {"dq": 5})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--shapes", nargs="+", default=["512x512", "128x1024", "256x1024", "1024x1024", "24x2048", "64x2048", "128x2048", "16x4096", "32x4096", "8x8192"])
    args = ap.parse_args()
    d = 128
    print(f"causal={args.causal}: backward ms, median of {args.rounds} rounds x {args.iters} calls")
    print("| bh x N | row tiles | " + " | ".join(n for n, _ in VARIANTS) + " |")
    print("|---|---|" + "---|" * len(VARIANTS))
    for shp in args.shapes:
        bh, n = (int(x) for x in shp.split("x"))
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(4))
        o, lse = ext.forward(q, k, v, args.causal, d ** -0.5, 64, 128)
        res = {name: [] for name, _ in VARIANTS}
        for rnd in range(args.rounds + 1):
            for name, opts in VARIANTS:
                for key in ("dkdv", "dq", "dq_w4"):
                    ext.set_option(key, opts.get(key, 0))
                try:
                    ext.backward(q, k, v, o, do, lse, args.causal, d ** -0.5, 64, 128)
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(args.iters):
                        ext.backward(q, k, v, o, do, lse, args.causal, d ** -0.5, 64, 128)
                    b.record()
                    torch.cuda.synchronize()
                    if rnd:
                        res[name].append(a.elapsed_time(b) / args.iters)
                except RuntimeError:
                    res[name].append(float("nan"))
        for key in ("dkdv", "dq", "dq_w4"):
            ext.set_option(key, 0)
        print(f"| {bh} x {n} | {bh * ((n + 255) // 256)} | " + " | ".join("%.3f" % statistics.median(res[name]) for name, _ in VARIANTS) + " |", flush=True)
        del q, k, v, do, o, lse
        ext.release_workspace()


if __name__ == "__main__":
    main()
