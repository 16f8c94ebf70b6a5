"""fa_ex_backward with Nq != Nk under the (shifted) causal diagonal: the dS hand-over (default) against the recomputing pass
(option dq = 5), interleaved, ms per call.

    python tools/ex_causal_handover.py
"""
import sys, statistics
sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext
for bh, nq, nk in ((32, 2048, 4096), (64, 4096, 8192), (128, 1024, 2048)):
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn((bh, nq, 128), device="cuda", dtype=torch.bfloat16, generator=g); do = torch.randn_like(q)
    k = torch.randn((bh, nk, 128), device="cuda", dtype=torch.bfloat16, generator=g); v = torch.randn_like(k)
    o, lse = ext.ex_forward(q, k, v, True, 0.088)
    res = {}
    for rnd in range(6):
        for name, dq in (("hand-over (default)", 0), ("recompute", 5)):
            ext.set_option("dq", dq)
            ext.ex_backward(q, k, v, o, do, lse, True, 0.088); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): ext.ex_backward(q, k, v, o, do, lse, True, 0.088)
            b.record(); torch.cuda.synchronize()
            if rnd: res.setdefault(name, []).append(a.elapsed_time(b) / 5)
    ext.set_option("dq", 0)
    print(bh, nq, nk, {k_: round(statistics.median(v_), 3) for k_, v_ in res.items()})
