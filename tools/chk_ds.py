"""dS hand-over backward (option dq = 6) against the recomputing split backward (dq = 5): equal up to the rounding of differently
ordered sums (the row constants come from the preparation launch there, from the dQ kernel here).  Shapes cover ragged N, the causal diagonal, several tiles."""
import sys
sys.path.insert(0, "flashattention-pytorch_amd")
import torch, flashattention_lab_cuda as ext
torch.manual_seed(0)
bad = 0
for dt in (torch.bfloat16, torch.float16):
    for bh, n in ((2, 256), (3, 300), (2, 1024), (1, 1000), (4, 2048), (2, 4096 + 77)):
        for causal in (False, True):
            d = 128
            q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=dt) for _ in range(4))
            o, lse = ext.forward(q, k, v, causal, d ** -0.5, 64, 128)
            ext.set_option("dq", 5); ext.set_option("dkdv", 5)
            ref = ext.backward(q, k, v, o, do, lse, causal, d ** -0.5, 64, 128)
            ext.set_option("dkdv", 0); ext.set_option("dq", 6)
            got = ext.backward(q, k, v, o, do, lse, causal, d ** -0.5, 64, 128)
            ext.set_option("dq", 0)
            diffs = [float((a.float() - b.float()).abs().max()) for a, b in zip(got, ref)]
            eq = [bool(torch.equal(a, b)) for a, b in zip(got, ref)]
            scale = [float(t.float().abs().max()) for t in ref]
            ok = all(df <= 2e-2 * sc for df, sc in zip(diffs, scale)) and all(bool(torch.isfinite(t.float()).all()) for t in got)
            bad += not ok
            print(dt, bh, n, causal, "max diff dq/dk/dv %.2e %.2e %.2e (max abs %.2e %.2e %.2e)" % (*diffs, *scale), "bitwise", eq, "OK" if ok else "FAIL", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
