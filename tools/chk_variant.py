import sys
sys.path.insert(0, "flashattention-pytorch_amd")
import torch, flashattention_lab_cuda as ext
opt, vals = sys.argv[1], [int(x) for x in sys.argv[2:]]
torch.manual_seed(0)
bh, n, d = 8, 1024, 128
q, k, v, do = (torch.randn((bh, n, d), device="cuda", dtype=torch.bfloat16) for _ in range(4))
o, lse = ext.forward(q, k, v, False, d ** -0.5, 64, 128)
ref = ext.backward(q, k, v, o, do, lse, False, d ** -0.5, 64, 128)
for val in vals:
    ext.set_option(opt, val)
    got = ext.backward(q, k, v, o, do, lse, False, d ** -0.5, 64, 128)
    ext.set_option(opt, 0)
    print(opt, val, [float((a.float() - b.float()).abs().max()) for a, b in zip(got, ref)], [bool(torch.equal(a, b)) for a, b in zip(got, ref)])
