"""Which kernels serve small launches: forward / backward time with the 4-wave / 128-row kernels forced (option small_grid = 2) and
with 256-row tiles forced (small_grid = 1), around the limits of csrc/fa_fwd_mfma.hip: small_grid() (profiles/r02_small_launches.md).

    python tools/small_launch_sweep.py [--head-dim 128]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flashattention-pytorch_amd"))
import torch
import flashattention_lab_cuda as ext


def timed(fn, iters=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--head-dim", type=int, default=128)
    args = ap.parse_args()
    d = args.head_dim
    variants = [("default", {}), ("4-wave", {"small_grid": 2}), ("256-row", {"small_grid": 1})]
    for causal in (False, True):
        print(f"d = {d}, causal = {causal}")
        print("| bh x N | tiles | " + " | ".join("fwd " + n for n, _ in variants) + " | " + " | ".join("bwd " + n for n, _ in variants) + " |")
        print("|---|---|" + "---|" * (2 * len(variants)))
        for bh, n in ((8, 512), (8, 2048), (12, 2048), (16, 2048), (20, 2048), (24, 2048), (28, 2048), (32, 2048), (40, 2048), (48, 2048), (64, 2048),
                      (4, 8192), (6, 8192), (8, 8192), (12, 8192), (16, 8192)):
            q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16) for _ in range(4))
            o, lse = ext.forward(q, k, v, causal, d ** -0.5, 64, 128)
            cf, cb = [], []
            for _, opts in variants:
                for key, val in opts.items():
                    ext.set_option(key, val)
                cf.append("%.3f" % timed(lambda: ext.forward(q, k, v, causal, d ** -0.5, 64, 128)))
                cb.append("%.3f" % timed(lambda: ext.backward(q, k, v, o, do, lse, causal, d ** -0.5, 64, 128)))
                for key in opts:
                    ext.set_option(key, 0)
            print(f"| {bh} x {n} | {bh * ((n + 255) // 256)} | " + " | ".join(cf) + " | " + " | ".join(cb) + " |", flush=True)


if __name__ == "__main__":
    main()
