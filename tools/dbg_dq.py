import sys
sys.path.insert(0, "flashattention-pytorch_amd"); sys.path.insert(0, ".")
import torch
import flashattention_lab_cuda as ext
from oracle import attention_oracle as orc
from tests.helpers import make_qkv
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
for n, causal in ((64, False), (128, False), (256, False), (512, False), (300, False), (512, True)):
    q, k, v, do = make_qkv(1, n, 128, torch.bfloat16, seed=1)
    rq, rk, rv, ro, rlse = orc.exact_attention_backward(q, k, v, do, causal, 128 ** -0.5, math_dtype=torch.float64)
    qd, kd, vd, dod = (t.cuda() for t in (q, k, v, do))
    o, lse = ext.forward(qd, kd, vd, causal, 128 ** -0.5, 64, 128)
    ext.set_option("small_grid", 1); ext.set_option("dq", 5)
    dq, dk, dv = ext.backward(qd, kd, vd, o, dod, lse, causal, 128 ** -0.5, 64, 128)
    ext.set_option("dq", 0); ext.set_option("small_grid", 0)
    err = (dq.cpu().float() - rq.float()).abs()[0]
    nb = (n + 31) // 32
    e = torch.zeros(nb, 4)
    for i in range(nb):
        for j in range(4):
            e[i, j] = err[32 * i:32 * i + 32, 32 * j:32 * j + 32].max()
    print(f"N={n} causal={causal} max|dq err| per (32-row block, 32-col block); |dq|max={rq.abs().max():.3f}")
    print(e)
    print("dk err", (dk.cpu().float() - rk.float()).abs().max().item(), "dv err", (dv.cpu().float() - rv.float()).abs().max().item())
