"""Forward / backward time of the extended attention path (fa_ex_forward / fa_ex_backward) per feature and per kernel
family (16-bit MFMA kernels, csrc/fa_ex_mfma.hip; exact-f32 kernels, csrc/fa_ex.hip), with the visible fraction of the
score matrix accounted for (algorithmic FLOPs = 4 / 10 x visible (q, key) pairs x d).

    python tools/bench_ex.py [--bh 32] [--nq 2048] [--nk 4096] [--head-dim 128] [--dtype bf16] [--paths default,mfma,exact]
"""
import argparse
import json
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext


def timed(fn, iters=5):
    for _ in range(2):
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
    ap.add_argument("--bh", type=int, default=32)
    ap.add_argument("--nq", type=int, default=2048)
    ap.add_argument("--nk", type=int, default=4096)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--paths", default="mfma,exact")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    bh, nq, nk, d = args.bh, args.nq, args.nk, args.head_dim
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn((bh, nq, d), device="cuda", dtype=dt, generator=g)
    k, v = (torch.randn((bh, nk, d), device="cuda", dtype=dt, generator=g) for _ in range(2))
    do = torch.randn((bh, nq, d), device="cuda", dtype=dt, generator=g)
    qi, kj = torch.arange(nq, device="cuda").unsqueeze(1), torch.arange(nk, device="cuda").unsqueeze(0)
    dense = (torch.rand((nq, nk), device="cuda", generator=g) < 0.5).to(torch.uint8)
    dense[:, 0] = 1
    bm = (torch.rand(((nq + 127) // 128, (nk + 127) // 128), device="cuda", generator=g) < 0.25).to(torch.uint8)
    bm[:, 0] = 1
    cases = {
        "plain (Nq != Nk)": dict(),
        "causal, bottom-right aligned": dict(causal=True),
        "dense mask, half the pairs": dict(mask=dense),
        "the causal mask handed over as a dense one": dict(mask=(kj <= qi + (nk - nq)).to(torch.uint8)),
        "block-sparse 128x128, a quarter of the tiles": dict(block_mask=bm, br=128, bc=128),
        "dropout 0.1": dict(dropout_p=0.1, seed=1),
        "causal + dropout 0.1": dict(causal=True, dropout_p=0.1, seed=1),
    }
    rows = []
    paths = ["exact"] if args.dtype == "fp32" else args.paths.split(",")
    for path in paths:
      ext.set_option("ex_path", {"mfma": 3, "exact": 1}.get(path, 0))   # 3: the extended MFMA kernels even where the plain ones would do; default: the library's own choice
      for name, kw in cases.items():
        kw = dict(kw)
        causal = kw.pop("causal", False)
        vis = torch.ones((nq, nk), dtype=torch.bool, device="cuda")
        if causal:
            vis &= kj <= qi + (nk - nq)
        if "mask" in kw:
            vis &= kw["mask"] != 0
        if "block_mask" in kw:
            vis &= kw["block_mask"].repeat_interleave(128, 0)[:nq].repeat_interleave(128, 1)[:, :nk] != 0
        frac = vis.float().mean().item()
        o, lse = ext.ex_forward(q, k, v, causal, d ** -0.5, **kw)
        tf = timed(lambda: ext.ex_forward(q, k, v, causal, d ** -0.5, **kw), args.iters)
        tb = timed(lambda: ext.ex_backward(q, k, v, o, do, lse, causal, d ** -0.5, **kw), args.iters)
        pairs = bh * nq * nk * frac
        rows.append(dict(kernels=path, case=name, visible_fraction=round(frac, 4), fwd_ms=round(tf, 3), bwd_ms=round(tb, 3),
                         fwd_tflops=round(4 * pairs * d / tf / 1e9, 2), bwd_tflops=round(10 * pairs * d / tb / 1e9, 2)))
    ext.set_option("ex_path", 0)
    print(json.dumps(dict(shape=dict(bh=bh, nq=nq, nk=nk, d=d, dtype=args.dtype),
                          kernels="mfma: v_mfma_f32_32x32x16 (peak 2500 TFLOP/s); exact: v_mfma_f32_16x16x4_f32 (peak 157 TFLOP/s)",
                          rows=rows), indent=1))


if __name__ == "__main__":
    main()
