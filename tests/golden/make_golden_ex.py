"""Golden vectors for the extended attention path, produced by running the REFERENCE's own code:
`MultiHeadAttention._block_sparse_flash_attention` and `look_ahead_mask_` of
/root/reference/src/fa3/torch/flashattention_pytorch.py (:94-174, :176-190).

That module cannot be imported as a whole — its first lines import `tiktoken` (not installed; there is no network) and its
tail downloads a dataset — so the two definitions are taken out of its syntax tree (`ast`) and executed on their own, in
this container only.  Nothing is written to the reference tree and no reference source is stored here: each .npz holds
inputs (q, k, v, masks) and the reference's output o.  Dropout is off (p = 0, eval mode): the reference draws its mask from
torch's global generator, which no kernel can reproduce; the dropout cases are pinned by the oracle's restatement of the
counter-based generator instead (oracle/attention_oracle.py: dropout_keep).

One more defect of the reference shapes the cases: inside a tile, a row whose keys are ALL masked gets
`exp(-inf - (-inf)) = NaN` (:144-145) and poisons the row — so the causal cases use one key tile (block_size >= Nk) or a
block-sparse mask that skips the tiles above the diagonal, and every dense-mask row keeps a key.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ex.py
"""
import ast
import json
import math
import os

import numpy as np
import torch
from torch import nn

SRC = "/root/reference/src/fa3/torch/flashattention_pytorch.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_definitions():
    tree = ast.parse(open(SRC).read())
    keep = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == "MultiHeadAttention")
            or (isinstance(n, ast.FunctionDef) and n.name == "look_ahead_mask_")]
    assert len(keep) == 2
    ns = {"torch": torch, "nn": nn, "math": math}
    exec(compile(ast.Module(body=keep, type_ignores=[]), SRC, "exec"), ns)
    return ns["MultiHeadAttention"], ns["look_ahead_mask_"]


def main():
    MHA, look_ahead = load_reference_definitions()
    cases = [
        # tag, seed, B, H, Nq, Nk, d, block, tau, causal (look_ahead_mask_), dense mask density, block-sparse density
        ("ex_causal_offset_kv_longer", 301, 1, 2, 40, 72, 32, 128, 1.0, True, None, None),
        ("ex_causal_offset_square_tau", 302, 2, 2, 96, 96, 64, 128, 0.7, True, None, None),
        ("ex_dense_mask", 303, 1, 3, 50, 70, 48, 128, 1.0, False, 0.6, None),
        ("ex_block_sparse", 304, 1, 2, 96, 160, 64, 32, 1.0, False, None, 0.5),
        ("ex_block_sparse_causal", 305, 1, 2, 128, 128, 128, 32, 1.3, True, None, 0.6),
        ("ex_cross_len_plain", 306, 2, 1, 33, 150, 40, 64, 1.0, False, None, None),
    ]
    for tag, seed, b, h, nq, nk, d, blk, tau, causal, dens, bdens in cases:
        g = torch.Generator().manual_seed(seed)
        q = torch.randn((b, h, nq, d), generator=g)
        k = torch.randn((b, h, nk, d), generator=g)
        v = torch.randn((b, h, nk, d), generator=g)
        mask = None
        if causal:
            mask = look_ahead(nq, nk)
        if dens is not None:
            m = torch.rand((1, 1, nq, nk), generator=g) < dens
            m[..., 0] = True                                  # every row keeps a key (an all -inf row is NaN in the reference)
            mask = m if mask is None else (mask & m)
        br, bc = min(blk, nq), min(blk, nk)
        nbr, nbc = math.ceil(nq / br), math.ceil(nk / bc)
        bmask = torch.ones((nbr, nbc), dtype=torch.int64)
        if bdens is not None:
            bmask = (torch.rand((nbr, nbc), generator=g) < bdens).long()
            bmask[:, 0] = 1                                   # every row block keeps a tile (and, under the causal mask, a visible key)
            if causal:                                        # see the docstring: no tile may hold a fully masked row
                bmask = torch.tril(bmask)
        mha = MHA(d_model=h * d, num_heads=h, dropout=0.0, use_fused_qkv=False, block_size=blk).eval()
        with torch.no_grad():
            o = mha._block_sparse_flash_attention(q, k, v, tau=tau, mask=mask, block_sparse_mask=bmask)
        assert torch.isfinite(o).all(), tag
        meta = dict(tag=tag, seed=seed, b=b, h=h, nq=nq, nk=nk, d=d, block_size=blk, br=br, bc=bc, tau=tau, causal=bool(causal),
                    dense_mask=dens is not None, block_sparse=bdens is not None, dtype="fp32",
                    source="MultiHeadAttention._block_sparse_flash_attention, flashattention_pytorch.py:94-174")
        store = dict(q=q.numpy(), k=k.numpy(), v=v.numpy(), o=o.numpy(), block_mask=bmask.numpy().astype(np.uint8),
                     meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        if mask is not None:
            store["mask"] = mask.reshape(nq, nk).numpy().astype(np.uint8)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **store)
        print(tag, tuple(o.shape), float(o.abs().max()))


if __name__ == "__main__":
    main()
