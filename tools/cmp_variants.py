"""Compare the forward output of a kernel variant (options on the command line: name=value) with the default build."""
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("--"))
causal = "--causal" in sys.argv
bh, n, d = 8, 4096, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
o0, l0 = ext.forward(q, k, v, causal, d ** -0.5, 64, 128)
for kk, vv in opts.items():
    ext.set_option(kk, int(vv))
o1, l1 = ext.forward(q, k, v, causal, d ** -0.5, 64, 128)
torch.cuda.synchronize()
do = (o0.float() - o1.float()).abs()
dl = (l0 - l1).abs()
print("max |do|", do.max().item(), "max |dlse|", dl.max().item(), "nan", torch.isnan(o1.float()).sum().item())
rows = do.amax(dim=-1)          # (bh, n)
bad = (rows > 2e-2).nonzero()
print("rows with |do| > 2e-2:", len(bad), bad[:10].tolist(), bad[-5:].tolist())
worst = rows.flatten().topk(5)
print("worst rows", [(int(i) // n, int(i) % n, float(x)) for x, i in zip(worst.values, worst.indices)])
