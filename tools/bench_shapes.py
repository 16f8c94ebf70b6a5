"""Forward / backward time of the C-ABI path over a list of shapes (bf16), HIP-event timed through the library's profiler.

    python tools/bench_shapes.py --shapes 256x4096x128 256x4096x96 256x4096x40 [--causal]
"""
import argparse
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["256x4096x128", "256x4096x96", "256x4096x64", "256x4096x40"])
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    print("| bh x N x d | causal | fwd ms | bwd ms | fwd+bwd TFLOP/s (14 N^2 d) |")
    print("|---|---|---|---|---|")
    for shp in args.shapes:
        bh, n, d = (int(x) for x in shp.split("x"))
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=dt, generator=g) for _ in range(4))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for it in range(2):   # warm-up, then timed
            ev[0].record()
            for _ in range(args.iters):
                o, lse = ext.forward(q, k, v, args.causal, d ** -0.5, 64, 128)
            ev[1].record()
            for _ in range(args.iters):
                ext.backward(q, k, v, o, do, lse, args.causal, d ** -0.5, 64, 128)
            ev[2].record()
            torch.cuda.synchronize()
        f, b = ev[0].elapsed_time(ev[1]) / args.iters, ev[1].elapsed_time(ev[2]) / args.iters
        flops = 14.0 * bh * n * n * d * (0.5 if args.causal else 1.0)
        print(f"| {bh} x {n} x {d} | {args.causal} | {f:.3f} | {b:.3f} | {flops / (f + b) * 1e-9:.0f} |", flush=True)


if __name__ == "__main__":
    main()
