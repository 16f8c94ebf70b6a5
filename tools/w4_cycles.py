"""Shader-clock cycles per (32-query x 64-key) block of the one-wave-per-SIMD dK/dV kernel, per ablation variant
(option dkdv_abl with bit 5 set: the kernel stamps its block loop and leaves (cycles, blocks) in dk[0..7]).

    python tools/w4_cycles.py [--abl 32 33 34 36 40 59 63]
"""
import argparse
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

NAMES = {32: "full", 33: "no vector slices", 34: "no DMA / wait / barrier", 36: "no operand requests", 40: "no row constants",
         59: "requests + MFMAs only", 63: "MFMAs only"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="dkdv", choices=["dkdv", "dq"])
    ap.add_argument("--abl", type=int, nargs="+", default=None)
    ap.add_argument("--bh", type=int, default=256)
    ap.add_argument("--seqlen", type=int, default=4096)
    args = ap.parse_args()
    dq_mode = args.kernel == "dq"
    if args.abl is None:
        args.abl = [32, 33, 34, 36, 39] if dq_mode else [32, 33, 34, 36, 40, 59, 63]
    opt, abl_opt, prof_name = ("dq", "dq_abl", "bwd_dq_mfma") if dq_mode else ("dkdv", "dkdv_abl", "bwd_mfma")
    mfmas = 96 if dq_mode else 64    # per stamped unit: a 64-key tile (dQ) / a 32-query block (dK/dV)
    NAMES.update({39: "MFMAs only"} if dq_mode else {})
    d, n, bh = 128, args.seqlen, args.bh
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(4))
    o, lse = ext.forward(q, k, v, False, d ** -0.5, 64, 128)
    ext.set_option(opt, 5)
    print(f"{args.kernel} stream kernel, bh={bh} N={n} d={d}: {mfmas} MFMAs per stamped unit")
    print("| variant | cycles / unit | cycles / MFMA | kernel ms | clock GHz (cycles x 16 WG per CU / time; loop only) |")
    print("|---|---|---|---|---|")
    try:
        for abl in args.abl:
            ext.set_option(abl_opt, abl)
            for _ in range(2):
                ext.backward(q, k, v, o, do, lse, False, d ** -0.5, 64, 128)
            ext.profile_enable(True)
            for _ in range(3):
                dq, dk, dv = ext.backward(q, k, v, o, do, lse, False, d ** -0.5, 64, 128)
            torch.cuda.synchronize()
            ms = ext.profile_report()[prof_name]
            ext.profile_enable(False)
            ms = ms[1] / ms[0]
            st = (dq if dq_mode else dk).view(torch.int32).flatten()[:2].cpu().tolist()
            cyc = st[0] / max(st[1], 1)
            wg_per_cu = bh * ((n + 255) // 256) / 256.0
            print(f"| {NAMES.get(abl, abl)} | {cyc:.0f} | {cyc / mfmas:.1f} | {ms:.3f} | {st[0] * wg_per_cu / (ms * 1e-3) / 1e9:.2f} |")
    finally:
        ext.set_option(abl_opt, 0)
        ext.set_option(opt, 0)


if __name__ == "__main__":
    main()
