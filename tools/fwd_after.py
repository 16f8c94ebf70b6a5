"""Why does the unchanged forward run slower behind the dQ product kernel?  One process: [forward, backward] steps with the recomputing
backward (dq = 5) and with the dS hand-over (default), optionally with an idle spin (torch.cuda._sleep) or a big memset between
backward and forward; per-kernel HIP-event times from the library."""
import sys, statistics
sys.path.insert(0, "flashattention-pytorch_amd")
import torch, flashattention_lab_cuda as ext
bh, n, d = 256, 4096, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(4))
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
def run(dq, between, rounds=6):
    ext.set_option("dq", dq)
    res = {}
    for r in range(rounds + 1):
        ext.profile_enable(True)
        for _ in range(3):
            o, lse = ext.forward(q, k, v, False, d ** -0.5, 64, 128)
            ext.backward(q, k, v, o, do, lse, False, d ** -0.5, 64, 128)
            if between == "sleep": torch.cuda._sleep(2_000_000)       # ~1 ms of an idle chip
            elif between == "memset": junk.zero_()                      # 1 GiB of writes: sweeps L2 / MALL
        torch.cuda.synchronize()
        prof = ext.profile_report(); ext.profile_enable(False)
        if r:
            for kname, (c, ms) in prof.items(): res.setdefault(kname, []).append(ms / c)
    ext.set_option("dq", 0)
    return {kname: round(statistics.median(x), 3) for kname, x in res.items()}
for rep in range(2):
    for dq in (5, 0):
        for between in ("none", "sleep", "memset"):
            print("recompute" if dq == 5 else "hand-over", between, run(dq, between), flush=True)
