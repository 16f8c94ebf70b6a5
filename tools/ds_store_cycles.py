"""Shader-clock stamps of the dK/dV stream kernel's block loop (option dkdv_abl: 32 stamps, +64 no dS stores, +512 one tile
per workgroup, +2048 no mask branch in the loop): cycles per block by row length, and — for the LAST key tile of workgroup 0
under the causal mask (8 blocks, all on the diagonal) — every wave's own time per block and the time between barriers.

    python tools/ds_store_cycles.py
"""
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

d, bh = 128, 128
g = torch.Generator(device="cuda").manual_seed(0)
print("| N | causal | variant | cycles | blocks | cycles / block |")
print("|---|---|---|---|---|---|")
detail = []
for n in (256, 1024, 4096):
    q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(4))
    for causal in (False, True):
        o, lse = ext.forward(q, k, v, causal, d ** -0.5, 64, 128)
        ext.set_option("dq", 6)
        variants = (("ds", 32), ("ds no stores", 96)) + ((("ds, 1 tile / wg", 544), ("ds no stores, 1 tile / wg", 608), ("ds, no mask branch in the loop (wrong results)", 2080)) if causal else ())
        for name, abl in variants:
            ext.set_option("dkdv_abl", abl)
            for _ in range(3):
                dq_, dk, dv = ext.backward(q, k, v, o, do, lse, causal, d ** -0.5, 64, 128)
            torch.cuda.synchronize()
            st = dk.view(torch.int32).flatten()[:32 + 256 + 320].cpu().tolist()
            print(f"| {n} | {causal} | {name} | {st[0]} | {st[1]} | {st[0] / max(st[1], 1):.0f} |")
            if causal and n == 1024 and abl in (32, 96):   # the stamped tile is workgroup 0's second one: keys 768 .. 1023, 8 blocks
                t = [[x & 0xffffffff for x in st[32 + 64 * w: 32 + 64 * w + 16]] for w in range(4)]
                t0 = min(x for row in t for x in row if x)
                detail.append(f"\n{name}: per wave and block, cycles since the tile's first stamp: own stream done / behind the barrier")
                for w in range(4):
                    fbw = 2 * w
                    own = [t[w][2 * b] - (t[w][2 * b - 1] if b > fbw else t[0][2 * b - 1] if b else t[w][2 * b]) for b in range(fbw, 8)]
                    detail.append(f"  wave {w}: own stream time of blocks {fbw} .. 7 (from the barrier before): " + " ".join(str(x) for x in own))
        ext.set_option("dkdv_abl", 0)
        ext.set_option("dq", 0)
print("\n".join(detail))
