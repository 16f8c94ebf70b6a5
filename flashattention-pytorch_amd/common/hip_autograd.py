"""Shared machinery behind `fa{1,2,3}/cuda/impl.py`: extension lookup, (B,H) reshapes and the autograd Function
factory.  The three reference wrappers (/root/reference/src/fa{1,2,3}/cuda/impl.py) are copies of one another that
differ only in the extension attribute names and in FA3's two extra scalars; here that is one parametrised builder.
"""
from __future__ import annotations

from importlib import import_module
from typing import Optional, Tuple

import torch

_EXT_NAMES = ("flashattention_lab_cuda", "flashattention_lab._C")   # lookup order of the reference, impl.py:10
_ext_module = None


def load_extension():
    """Cached lookup of the native extension module by name; ImportError (with the reasons) if none imports."""
    global _ext_module
    if _ext_module is None:
        reasons = []
        for name in _EXT_NAMES:
            try:
                _ext_module = import_module(name)
                break
            except Exception as exc:  # noqa: BLE001 - any import-time failure means "not available"
                reasons.append(f"{name}: {exc}")
        else:
            raise ImportError("CUDA extension module not found (" + "; ".join(reasons) + ")")
    return _ext_module


def merge_bh(x: torch.Tensor) -> Tuple[torch.Tensor, Optional[Tuple[int, int]]]:
    """(B,H,N,d) -> ((BH,N,d) view, (B,H)); a 3-D tensor passes through with None."""
    if x.dim() == 3:
        return x, None
    b, h = x.shape[0], x.shape[1]
    return x.reshape(b * h, *x.shape[2:]), (b, h)


def split_bh(x: torch.Tensor, bh_shape):
    return x if bh_shape is None else x.reshape(*bh_shape, *x.shape[1:])


def make_attention_function(cls_name: str, fwd_attr: str, bwd_attr: str, n_extra: int = 0):
    """Build the `torch.autograd.Function` that fronts one pair of extension entry points.

    forward(ctx, q, k, v, causal, softmax_scale, br, bc, *extra) -> (o, lse); saves (q, k, v, o, lse), as the reference
    does (impl.py:40-58).  backward(ctx, do, dlse) ignores dlse (impl.py:61-73) and returns None for every scalar.
    `extra` are FA3's (stages, fp8), coerced to (int, bool).
    """
    coerce_extra = (int, bool)[:n_extra]

    def forward(ctx, q, k, v, causal, softmax_scale, br, bc, *extra):
        if len(extra) != n_extra:
            raise TypeError(f"{cls_name}.forward takes {n_extra} extra arguments, got {len(extra)}")
        if not (q.is_cuda and k.is_cuda and v.is_cuda):
            raise RuntimeError("Inputs must be CUDA tensors")
        ctx.scalars = (bool(causal), float(softmax_scale), int(br), int(bc)) + tuple(f(e) for f, e in zip(coerce_extra, extra))
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        o, lse = getattr(load_extension(), fwd_attr)(q, k, v, *ctx.scalars)
        ctx.save_for_backward(q, k, v, o, lse)
        return o, lse

    def backward(ctx, do, dlse):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = getattr(load_extension(), bwd_attr)(q, k, v, o, do.contiguous(), lse, *ctx.scalars)
        return (dq, dk, dv) + (None,) * (4 + n_extra)

    return type(cls_name, (torch.autograd.Function,), {"forward": staticmethod(forward), "backward": staticmethod(backward),
                                                       "__doc__": make_attention_function.__doc__})


def run_attention(fn_cls, q, k, v, *scalars):
    """Merge (B,H), apply the Function, split back: the body of every `fa?_cuda` (impl.py:75-85)."""
    qb, bh_shape = merge_bh(q)
    kb, _ = merge_bh(k)
    vb, _ = merge_bh(v)
    o, lse = fn_cls.apply(qb, kb, vb, *scalars)
    return split_bh(o, bh_shape), split_bh(lse, bh_shape)
