"""Forward-only timing of BASELINE config 5 (N=16384, d=128; B=1, H=16 as SURVEY §8d suggests): fa3 fp8=True — the all-e4m3
kernel (default) and the variant with the 16-bit P.V (option fp8_pv = 1) — against the bf16 forward, interleaved in one
process.  HIP-event kernel times per forward call from the library's profiler (quantisation launches included), and the
largest deviation of each fp8 output from the bf16 kernel's.  `--head-dim 64` times the round-trip path of the other head dims."""
import argparse
import json
import statistics
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--heads", type=int, default=16)
ap.add_argument("--seqlen", type=int, default=16384)
ap.add_argument("--head-dim", type=int, default=128)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
B, H, N, D = args.batch, args.heads, args.seqlen, args.head_dim
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((B * H, N, D), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
variants = [("bf16", False, 0), ("fp8", True, 0)] + ([("fp8_pv16", True, 1)] if D == 128 else [])
res = {}
CALLS = 10
for causal in (False, True):
    ref = None
    acc = {name: [] for name, _, _ in variants}
    kern = {}
    err = {}
    for rnd in range(args.rounds + 1):
        for name, fp8, pv in variants:
            ext.set_option("fp8_pv", pv)
            ext.profile_enable(True)
            for _ in range(CALLS):
                o, _ = ext.fa3_forward(q, k, v, causal, D ** -0.5, 64, 128, 2, fp8)
            torch.cuda.synchronize()
            prof = ext.profile_report()
            ext.profile_enable(False)
            if rnd == 0:
                if name == "bf16":
                    ref = o.float()
                else:
                    err[name] = (o.float() - ref).abs().max().item()
                continue
            acc[name].append(sum(ms for _c, ms in prof.values()) / CALLS)
            kern[name] = {kn: ms / CALLS for kn, (_c, ms) in prof.items()}
    ext.set_option("fp8_pv", 0)
    flops = 4.0 * B * H * N * N * D * ((N + 1) / (2 * N) if causal else 1.0)
    for name, _, _ in variants:
        tot = statistics.median(acc[name])
        res[f"{'causal' if causal else 'full'}_{name}"] = {"kernels_ms": kern[name], "total_ms": tot, "min_ms": min(acc[name]),
                                                          "tflops": flops / tot / 1e9, "max_abs_diff_vs_bf16_kernel": err.get(name)}
print(json.dumps({"config": f"fa3 forward B={B} H={H} N={N} d={D} bf16 tensors, median of {args.rounds} interleaved rounds x {CALLS} calls",
                  "results": res}, indent=1))
