"""Interleaved A/B of kernel variants in ONE process (cdna guide §5.4 rule 24): per round every variant runs the same
fwd+bwd; per-kernel HIP-event times are collected by the library's profiler; medians / minima over the rounds.

    python tools/ab.py [--rounds 7] [--causal] [--head-dim 128] [-v "name:opt=val,opt=val" ...]
"""
import argparse
import statistics
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

VARIANTS = [   # edit to taste: every key of ALL_KEYS is an fa_set_option name (csrc/fa_kernels.h)
    ("base", {}),
    ("fwd: lock-step kernel", {"fwd_stag": 2}),
    ("fwd: staggered kernel, 64-key tiles", {"fwd_stag": 1}),
]
ALL_KEYS = ["fwd_kb", "fwd_stag", "dkdv", "dq_kt", "fwd_rs", "dkdv_kreg", "fwd_eager", "fwd_hs", "fwd_tpw", "dq_tpw", "dkdv_tpw", "dq_nlf", "dq_w4", "fwd_abl", "small_grid", "fp8_rot", "dkdv_stg", "fwd_rd", "fwd_w2", "dkdv_abl", "dq", "dq_abl", "ds_chunk_mb"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=4096)
    ap.add_argument("--bh", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("-v", "--variant", action="append", default=[],
                    help='"name:opt=val,opt=val" (fa_set_option names); replaces the built-in list, "base" is always first')
    args = ap.parse_args()
    global VARIANTS
    if args.variant:
        VARIANTS = [("base", {})]
        for spec in args.variant:
            name, _, opts = spec.partition(":")
            VARIANTS.append((name, {kv.split("=")[0]: int(kv.split("=")[1]) for kv in opts.split(",") if kv}))
    keys = list(dict.fromkeys(ALL_KEYS + [k_ for _, o_ in VARIANTS for k_ in o_]))
    d, n, bh = args.head_dim, args.seqlen, args.bh
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float16, generator=g) for _ in range(4))
    res = {name: {} for name, _ in VARIANTS}
    for rnd in range(args.rounds + 1):
        for name, opts in VARIANTS:
            for key in keys:
                ext.set_option(key, opts.get(key, 0))
            ext.profile_enable(True)
            for _ in range(3):
                o, lse = ext.forward(q, k, v, args.causal, d ** -0.5, 64, 128)
                ext.backward(q, k, v, o, do, lse, args.causal, d ** -0.5, 64, 128)
            torch.cuda.synchronize()
            prof = ext.profile_report()
            ext.profile_enable(False)
            if rnd == 0:
                continue  # warm-up round
            for kname, (cnt, ms) in prof.items():
                res[name].setdefault(kname, []).append(ms / cnt)
    for key in keys:
        ext.set_option(key, 0)
    kernels = sorted({kn for r in res.values() for kn in r})
    print(f"bh={bh} N={n} d={d} causal={args.causal} rounds={args.rounds}: median (min) ms per launch")
    print("| variant | " + " | ".join(kernels) + " | sum |")
    print("|---|" + "---|" * (len(kernels) + 1))
    for name, _ in VARIANTS:
        cells, tot = [], 0.0
        for kn in kernels:
            xs = res[name].get(kn, [])
            if xs:
                cells.append(f"{statistics.median(xs):.3f} ({min(xs):.3f})")
                tot += statistics.median(xs)
            else:
                cells.append("-")
        print(f"| {name} | " + " | ".join(cells) + f" | {tot:.3f} |")


if __name__ == "__main__":
    main()
