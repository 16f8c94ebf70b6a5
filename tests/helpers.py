"""Shared helpers for the parity tests (golden loading, seeded inputs, tolerances)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def golden_tags(pattern="*"):
    """the hot path's fixtures (tests/golden/make_golden.py); the extended-attention ones are listed by ex_golden_tags()"""
    tags = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, pattern + ".npz")))
    return [t for t in tags if not t.startswith("ex_")] if pattern == "*" else tags


def ex_golden_tags():
    return golden_tags("ex_*")


def load_ex_golden(tag):
    """(meta, tensors) of an extended-attention fixture (tests/golden/make_golden_ex.py): q (B,H,Nq,d), k, v (B,H,Nk,d),
    o (B,H,Nq,d) fp32; block_mask uint8; mask uint8 (Nq, Nk) when present."""
    z = np.load(os.path.join(GOLDEN_DIR, tag + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: torch.from_numpy(z[k].copy()) for k in z.files if k != "meta"}


def load_golden(tag):
    z = np.load(os.path.join(GOLDEN_DIR, tag + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    dt = DT[meta["dtype"]]
    out = {}
    for key in z.files:
        if key == "meta":
            continue
        if key.endswith("_bits"):
            out[key[:-5]] = torch.from_numpy(z[key].copy()).view(dt)
        else:
            out[key] = torch.from_numpy(z[key].copy())
    return meta, out


def dtype_tolerances(dtype):
    # the reference's own bar: tests/utils.py:31-36 (fp16/bf16 5e-2, fp32 1e-4)
    if dtype in (torch.float16, torch.bfloat16):
        return {"rtol": 5e-2, "atol": 5e-2}
    return {"rtol": 1e-4, "atol": 1e-4}


def make_qkv(bh, n, d, dtype, seed, device="cpu", with_do=True):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ts = [torch.randn((bh, n, d), generator=g, dtype=torch.float32).to(dtype) for _ in range(4 if with_do else 3)]
    return [t.to(device) for t in ts]


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()
