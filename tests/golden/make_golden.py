"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python.

Run once in the build container (the reference tree does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from /root/reference/src (read-only, nothing is written there):
  common.correctness.reference_attention / reference_backward   (non-causal oracle)
  fa1.torch.impl.fa1_forward_torch / fa1_backward_torch          (causal + multi-tile oracle)
  fa1.op.fa1_attention(backend="torch")                          (BASELINE.json config 1)
  fa1/fa2/fa3.spec.pick_fa?_spec                                  (tile table)
See SURVEY.md §8(c) for why these and not the FA2/FA3-fp8 torch paths (defects D2-D7).

Each .npz holds INPUTS (q, k, v, do) and the reference's OUTPUTS (o, lse, dq, dk, dv) —
data only.  16-bit tensors are stored as raw int16 (`*_bits`) so they round-trip exactly.
"""
import json
import os
import sys

import numpy as np
import torch

REF_SRC = "/root/reference/src"
sys.dont_write_bytecode = True
sys.path.insert(0, REF_SRC)

from common.correctness import reference_attention, reference_backward  # noqa: E402
from fa1.torch.impl import fa1_backward_torch, fa1_forward_torch  # noqa: E402
from fa1.spec import pick_fa1_spec  # noqa: E402
from fa2.spec import pick_fa2_spec  # noqa: E402
from fa3.spec import pick_fa3_spec  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def pack(name, t, store):
    t = t.detach().contiguous()
    if t.dtype in (torch.float16, torch.bfloat16):
        store[name + "_bits"] = t.view(torch.int16).numpy()
    else:
        store[name] = t.numpy()


def run_case(tag, seed, bh, n, d, dtype, causal, spec, backward=True, gen="global"):
    dt = DT[dtype]
    if gen == "global":  # tests/utils.py:7-16 order: manual_seed, then q, k, v
        torch.manual_seed(seed)
        q = torch.randn((bh, n, d), dtype=dt)
        k = torch.randn((bh, n, d), dtype=dt)
        v = torch.randn((bh, n, d), dtype=dt)
    else:  # benchmarks/bench_utils.py:83-97: seeded Generator, order q, k, v
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        q = torch.randn((bh, n, d), dtype=dt, generator=g)
        k = torch.randn((bh, n, d), dtype=dt, generator=g)
        v = torch.randn((bh, n, d), dtype=dt, generator=g)
    scale = d ** -0.5
    store = {}
    meta = dict(tag=tag, seed=seed, bh=bh, n=n, d=d, dtype=dtype, causal=bool(causal), softmax_scale=scale,
                br=spec.br, bc=spec.bc, torch_version=torch.__version__)
    if causal:
        o, lse = fa1_forward_torch(q, k, v, True, scale, spec.br, spec.bc)
        meta["forward_source"] = "src/fa1/torch/impl.py:fa1_forward_torch"
    else:
        o, lse = reference_attention(q, k, v, causal=False, softmax_scale=scale)
        meta["forward_source"] = "src/common/correctness.py:reference_attention"
    for nm, t in (("q", q), ("k", k), ("v", v), ("o", o), ("lse", lse)):
        pack(nm, t, store)
    if backward:
        if gen == "global":
            do = torch.randn_like(o)  # tests/test_correctness_fa2.py:101
        else:
            do = torch.randn(o.shape, dtype=dt, generator=g)
        if causal:
            dq, dk, dv = fa1_backward_torch(q, k, v, o, do, lse, True, scale, spec.br, spec.bc)
            meta["backward_source"] = "src/fa1/torch/impl.py:fa1_backward_torch"
        else:
            dq, dk, dv, _, _ = reference_backward(q, k, v, do, False, scale)
            meta["backward_source"] = "src/common/correctness.py:reference_backward"
        for nm, t in (("do", do), ("dq", dq), ("dk", dk), ("dv", dv)):
            pack(nm, t, store)
    store["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, tag + ".npz")
    np.savez(path, **store)
    print(f"{tag}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    # (1) the reference's own test cases (tests/test_correctness_fa{1,2,3}.py), same seeds, CPU RNG
    for causal in (False, True):
        c = "c" if causal else "n"
        for dtype in ("fp16", "fp32"):
            run_case(f"fa1_fwd_s0_2x16x32_{dtype}_{c}", 0, 2, 16, 32, dtype, causal, pick_fa1_spec(32), backward=False)
            run_case(f"fa1_fwd_s0_2x33x64_{dtype}_{c}", 0, 2, 33, 64, dtype, causal, pick_fa1_spec(64), backward=False)
            run_case(f"fa2_fwd_s10_1x24x32_{dtype}_{c}", 10, 1, 24, 32, dtype, causal, pick_fa2_spec(32), backward=False)
            run_case(f"fa2_fwd_s10_4x33x64_{dtype}_{c}", 10, 4, 33, 64, dtype, causal, pick_fa2_spec(64), backward=False)
            run_case(f"fa3_fwd_s20_2x24x32_{dtype}_{c}", 20, 2, 24, 32, dtype, causal, pick_fa3_spec(32), backward=False)
        run_case(f"fa1_bwd_s1_2x12x32_fp32_{c}", 1, 2, 12, 32, "fp32", causal, pick_fa1_spec(32))
        run_case(f"fa2_bwd_s11_2x16x40_fp32_{c}", 11, 2, 16, 40, "fp32", causal, pick_fa2_spec(40))
        run_case(f"fa1_cuda_s3_2x24x64_fp16_{c}", 3, 2, 24, 64, "fp16", causal, pick_fa1_spec(64))
        run_case(f"fa2_cuda_s13_2x32x48_fp16_{c}", 13, 2, 32, 48, "fp16", causal, pick_fa2_spec(48))
        run_case(f"fa3_cuda_s22_2x32x32_fp16_{c}", 22, 2, 32, 32, "fp16", causal, pick_fa3_spec(32))
    run_case("fa1_cons_s4_2x24x32_fp16_c", 4, 2, 24, 32, "fp16", True, pick_fa1_spec(32), backward=False)
    run_case("fa2_cons_s14_2x28x32_fp16_c", 14, 2, 28, 32, "fp16", True, pick_fa2_spec(32), backward=False)
    run_case("fa3_cons_s23_2x20x32_fp16_c", 23, 2, 20, 32, "fp16", True, pick_fa3_spec(32), backward=False)

    # (2) BASELINE.json config 1: FA1 forward, B=2 H=4 N=128 d=64 fp32, bench generator seed 0
    run_case("config1_fa1_fwd_8x128x64_fp32_n", 0, 8, 128, 64, "fp32", False, pick_fa1_spec(64), backward=False, gen="bench")
    # ... and through the dispatcher exactly as benchmarks/bench_fa1.py would call it
    from fa1.op import fa1_attention
    g = torch.Generator(device="cpu"); g.manual_seed(0)
    q = torch.randn((2, 4, 128, 64), generator=g); k = torch.randn((2, 4, 128, 64), generator=g)
    v = torch.randn((2, 4, 128, 64), generator=g)
    o, lse = fa1_attention(q, k, v, causal=False, backend="torch")
    z = np.load(os.path.join(OUT, "config1_fa1_fwd_8x128x64_fp32_n.npz"))
    assert np.abs(o.reshape(8, 128, 64).numpy() - z["o"]).max() < 1e-5, "dispatcher path disagrees with oracle"

    # (3) gap fillers the reference never tests: multi-tile, d=128, bf16, br != bc, ragged N
    for causal in (False, True):
        c = "c" if causal else "n"
        for dtype in ("fp32", "bf16"):
            run_case(f"gap_s100_2x300x64_{dtype}_{c}", 100, 2, 300, 64, dtype, causal, pick_fa2_spec(64))
            b101 = 1 if dtype == "fp32" else 2
            run_case(f"gap_s101_{b101}x300x128_{dtype}_{c}", 101, b101, 300, 128, dtype, causal, pick_fa2_spec(128))
        run_case(f"gap_s102_1x512x128_bf16_{c}", 102, 1, 512, 128, "bf16", causal, pick_fa2_spec(128))
        run_case(f"gap_s103_1x257x256_fp16_{c}", 103, 1, 257, 256, "fp16", causal, pick_fa2_spec(256))


if __name__ == "__main__":
    main()
