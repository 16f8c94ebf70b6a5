"""dS hand-over chunk sweep: whole-step time of fwd+bwd at each `ds_chunk_mb`, interleaved in ONE
process (cdna guide §5.4 rule 24), against the recomputing backward (dq = 5).  Step time by a pair of events around
each step on the launch stream; median and minimum over the rounds; backward workspace bytes per variant.

    python tools/sweep_ds_chunk.py [--rounds 7] [--causal] [--seqlen 4096] [--bh 256] [--chunks 256 512 1024 2048 4096 8800]
"""
import argparse
import statistics
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=10, help="steps per variant per round")
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--seqlen", type=int, default=4096)
    ap.add_argument("--bh", type=int, default=256)
    ap.add_argument("--chunks", type=int, nargs="+", default=[256, 512, 1024, 2048, 4096, 8800])
    args = ap.parse_args()
    d, n, bh = 128, args.seqlen, args.bh
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(4))
    variants = [("recompute (dq=5)", {"dq": 5})] + [(f"chunk {mb} MiB", {"ds_chunk_mb": mb, **({"dq": 6} if args.causal else {})}) for mb in args.chunks]   # (causal: the hand-over only by option)
    res = {name: [] for name, _ in variants}
    wsb = {}
    kern = {name: {} for name, _ in variants}
    for rnd in range(args.rounds + 1):
        for name, opts in variants:
            for key in ("dq", "ds_chunk_mb"):
                ext.set_option(key, opts.get(key, 0))
            ext.release_workspace()
            wsb[name] = int(ext._lib.fa_backward_workspace_bytes_fast(bh, n, d, 2, int(args.causal))) if opts.get("dq") != 5 \
                else int(ext._lib.fa_backward_workspace_bytes(bh, n, d, 2))
            for _ in range(2):
                o, lse = ext.forward(q, k, v, args.causal, d ** -0.5, 64, 128)
                ext.backward(q, k, v, o, do, lse, args.causal, d ** -0.5, 64, 128)
            torch.cuda.synchronize()
            ext.profile_enable(rnd == args.rounds)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.steps):
                o, lse = ext.forward(q, k, v, args.causal, d ** -0.5, 64, 128)
                ext.backward(q, k, v, o, do, lse, args.causal, d ** -0.5, 64, 128)
            b.record()
            torch.cuda.synchronize()
            if rnd == args.rounds:
                kern[name] = {kn: ms / args.steps for kn, (c, ms) in ext.profile_report().items()}
                ext.profile_enable(False)
            if rnd:
                res[name].append(a.elapsed_time(b) / args.steps)
    for key in ("dq", "ds_chunk_mb"):
        ext.set_option(key, 0)
    flops = 14.0 * bh * n * n * d * ((n + 1) / (2.0 * n) if args.causal else 1.0)
    print(f"bh={bh} N={n} d={d} causal={args.causal} rounds={args.rounds} x {args.steps} steps: fwd+bwd ms per step, median (min)")
    print("| variant | workspace GiB | ms / step | TFLOP/s (median) | kernel ms per step (last round): fwd / prep / dK,dV / dQ |")
    print("|---|---|---|---|---|")
    for name, _ in variants:
        xs = res[name]
        kk = kern[name]
        print(f"| {name} | {wsb[name] / 2**30:.3f} | {statistics.median(xs):.3f} ({min(xs):.3f}) | {flops / statistics.median(xs) / 1e9:.0f} | "
              f"{kk.get('fwd_mfma', 0):.3f} / {kk.get('bwd_delta', 0):.3f} / {kk.get('bwd_mfma', 0):.3f} / {kk.get('bwd_dq_mfma', 0):.3f} |")


if __name__ == "__main__":
    main()
