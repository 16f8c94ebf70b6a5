"""Host-side counterpart of the attention core of the reference's notebook model
(`MultiHeadAttention._block_sparse_flash_attention(q, k, v, tau, mask, block_sparse_mask)`,
/root/reference/src/fa3/torch/flashattention_pytorch.py:94-174, and its dense branch :80-87) over the HIP library's
extended entry points (`fa_ex_forward` / `fa_ex_backward`, include/fa_mi355x.h):

    flash_attention_ex(q, k, v, tau=1.0, mask=None, block_sparse_mask=None, block_size=128,
                       causal=False, dropout_p=0.0, seed=0, softmax_scale=None) -> o

q: (B, H, Nq, d) or (BH, Nq, d); k, v: (..., Nk, d).  `mask` follows the model's convention — a boolean / 0-1 tensor
broadcastable to (B, H, Nq, Nk), True / 1 = allowed (`look_ahead_mask_` builds the causal one for Nq != Nk, :176-190;
`causal=True` is the same mask without materialising it) — and `block_sparse_mask[i, j] == 0` skips tile (i, j) of
`block_size` x `block_size` (Algorithm 5).  Differentiable (autograd Function; the backward recomputes P and regenerates
the dropout mask from the seed).  No CPU path: the tensors must live on the GPU.
"""
from __future__ import annotations

import math

import torch


def look_ahead_mask(q_len, k_len=None, device=None):
    """True = allowed: key j visible to query i iff j <= i + (k_len - q_len) (flashattention_pytorch.py:176-190)."""
    k_len = q_len if k_len is None else k_len
    qi = torch.arange(q_len, device=device).unsqueeze(1)
    kj = torch.arange(k_len, device=device).unsqueeze(0)
    return (kj <= qi + (k_len - q_len)).unsqueeze(0).unsqueeze(0)


def normalize_mask(mask, lead, nq, nk):
    """The model's `mask` (True / non-zero = allowed), "broadcastable to (B, H, Nq, Nk)" as the reference uses it
    (`scores.masked_fill(mask[:, :, i0:i1, j0:j1] == 0, -inf)`, flashattention_pytorch.py:139-141: ordinary broadcasting against
    the (B, H, Br, Bc) scores), in one of the two forms the library takes: (Nq, Nk) when every leading dim is 1 — one mask
    shared by all (b,h) — else (BH, Nq, Nk).  `lead` = q's leading dims, (B, H) or (BH,).  A key-padding mask (B, 1, 1, Nk)
    or (1, 1, 1, Nk) is expanded over the query rows like any other broadcast dim."""
    m = mask != 0
    if m.dim() > len(lead) + 2:
        if all(s == 1 for s in m.shape[: m.dim() - 2]):
            m = m.reshape(m.shape[-2:])
        elif len(lead) == 1 and math.prod(m.shape[:-2]) == lead[0]:
            m = m.reshape(lead[0], *m.shape[-2:])      # a (B, H, ., .) mask with already merged (BH, N, d) tensors
        else:
            raise RuntimeError(f"mask of shape {tuple(mask.shape)} does not broadcast to {(*lead, nq, nk)}")
    while m.dim() < len(lead) + 2:
        m = m.unsqueeze(0)
    shared = all(s == 1 for s in m.shape[:-2])
    try:
        m = torch.broadcast_to(m, ((1,) * len(lead) if shared else lead) + (nq, nk))
    except RuntimeError as exc:
        raise RuntimeError(f"mask of shape {tuple(mask.shape)} does not broadcast to {(*lead, nq, nk)}") from exc
    return m.reshape(nq, nk) if shared else m.reshape(-1, nq, nk)


class _FlashAttnExFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, causal, scale, mask, block_mask, br, bc, dropout_p, seed):
        import flashattention_lab_cuda as ext

        o, lse = ext.ex_forward(q, k, v, causal, scale, mask, block_mask, br, bc, dropout_p, seed)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.args = (causal, scale, mask, block_mask, br, bc, dropout_p, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        import flashattention_lab_cuda as ext

        q, k, v, o, lse = ctx.saved_tensors
        causal, scale, mask, block_mask, br, bc, dropout_p, seed = ctx.args
        dq, dk, dv = ext.ex_backward(q, k, v, o, do.contiguous(), lse, causal, scale, mask, block_mask, br, bc, dropout_p, seed)
        return (dq, dk, dv) + (None,) * 8


def flash_attention_ex(q, k, v, tau=1.0, mask=None, block_sparse_mask=None, block_size=128, causal=False, dropout_p=0.0,
                       seed=0, softmax_scale=None):
    if not q.is_cuda:
        raise RuntimeError("Inputs must be CUDA tensors")   # as the reference's wrappers (src/fa2/cuda/impl.py:44)
    four_d = q.dim() == 4
    if four_d:
        b, h, nq, d = q.shape
        nk = k.shape[2]
        q3, k3, v3 = q.reshape(b * h, nq, d), k.reshape(b * h, nk, d), v.reshape(b * h, nk, d)
    else:
        q3, k3, v3 = q, k, v
        (_, nq, d), nk = q.shape, k.shape[1]
    scale = (tau / math.sqrt(d)) if softmax_scale is None else float(softmax_scale) * tau   # :134
    m = None
    if mask is not None:
        m = normalize_mask(mask, tuple(q.shape[:-2]), nq, nk)
    br = bc = int(block_size)
    if block_sparse_mask is not None:
        br, bc = min(br, nq), min(bc, nk)         # Br = min(block_size, q_len), Bc = min(block_size, kv_len)  (:100-101)
    o = _FlashAttnExFn.apply(q3, k3, v3, bool(causal), scale, m, block_sparse_mask, br, bc, float(dropout_p), int(seed))
    return o.reshape(q.shape) if four_d else o
