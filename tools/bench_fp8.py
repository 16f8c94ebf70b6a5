"""Forward-only timing of BASELINE config 5 (N=16384, d=128; B=1, H=16 as SURVEY §8d suggests): fa3 fp8=True vs the
bf16 forward.  HIP-event kernel times from the library's profiler."""
import json
import sys

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

B, H, N, D = 1, 16, 16384, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((B * H, N, D), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
res = {}
for causal in (False, True):
    for fp8 in (False, True):
        for _ in range(3):
            ext.fa3_forward(q, k, v, causal, D ** -0.5, 64, 128, 2, fp8)
        torch.cuda.synchronize()
        ext.profile_enable(True)
        for _ in range(10):
            ext.fa3_forward(q, k, v, causal, D ** -0.5, 64, 128, 2, fp8)
        torch.cuda.synchronize()
        prof = ext.profile_report()
        ext.profile_enable(False)
        ms = {k_: v_[1] / v_[0] for k_, v_ in prof.items()}
        flops = 4.0 * B * H * N * N * D * ((N + 1) / (2 * N) if causal else 1.0)
        tot = sum(ms.values())
        res[f"{'causal' if causal else 'full'}_{'fp8' if fp8 else 'bf16'}"] = {"kernels_ms": ms, "total_ms": tot, "tflops": flops / tot / 1e9}
print(json.dumps({"config": f"fa3 forward B={B} H={H} N={N} d={D} bf16 tensors", "results": res}, indent=1))
