"""Phase timeline of the staggered forward kernel (debug): shader-clock stamps of waves 0 and 4 of workgroup 0.

    python tools/trace_stag.py [--causal]
Events per half-step: phase body end, barrier end.  Prints the first steps and the median body / barrier-wait lengths.
"""
import statistics
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

bh, n, d = 256, 4096, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
ext.set_option("fwd_stag", 3 if "--kb4" in sys.argv else 1)
abl = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--abl=")]
ext.set_option("fwd_abl", abl[0] if abl else 0)
print("ablation flags", abl)
for _ in range(2):
    ext.forward(q, k, v, "--causal" in sys.argv, d ** -0.5, 64, 128)
buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
ext.debug_trace_buffer(buf)
ext.forward(q, k, v, "--causal" in sys.argv, d ** -0.5, 64, 128)
torch.cuda.synchronize()
ext.debug_trace_buffer(None)
t = buf.cpu().tolist()
for half in (0, 1):
    ev = t[half * 2048: half * 2048 + 2048]
    cnt = ev[0]
    st = ev[1:cnt]
    t0 = st[0]
    rel = [x - t0 for x in st]
    body = [rel[i] - rel[i - 1] for i in range(2, len(rel), 2)]      # barrier end -> next body end
    wait = [rel[i] - rel[i - 1] for i in range(1, len(rel), 2)]      # body end -> barrier end
    print(f"wave {4 * half}: {cnt - 1} stamps; first 24 (cycles since first): {rel[:24]}")
    print(f"   body lengths   (alternating V / M phases): even-index median {statistics.median(body[0::2]):.0f}, odd-index median {statistics.median(body[1::2]):.0f}")
    print(f"   barrier waits: even-index median {statistics.median(wait[0::2]):.0f}, odd-index median {statistics.median(wait[1::2]):.0f}")
    print(f"   total {rel[-1]} cycles for {len(rel) // 2} half-steps = {rel[-1] / (len(rel) // 2):.0f} per half-step")
