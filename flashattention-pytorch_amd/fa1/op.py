"""FA1 dispatcher — counterpart of /root/reference/src/fa1/op.py:7-28.

`fa1_attention(q, k, v, causal=False, softmax_scale=None, backend="auto") -> (o, lse)`

Differences from the reference, all deliberate (DESIGN.md "Boundary"):
  * "cuda" is the MI355X HIP path (ROCm tensors report `is_cuda`); it is the only compute backend
    shipped.  `backend="auto"` selects it and NEVER swallows its exceptions (the reference's
    `except Exception: -> triton` at op.py:16-19 would hide a kernel fault).
  * "triton" is refused (out of scope by construction) and "torch" — the reference's CPU tile
    loops — is not part of the product: its restatement lives in `oracle/` as test infrastructure.
    Both raise NotImplementedError; an unknown backend raises ValueError as in the reference.
"""
from .cuda.impl import fa1_cuda
from .spec import pick_fa1_spec


def fa1_attention(q, k, v, causal=False, softmax_scale=None, backend="auto"):
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** -0.5
    spec = pick_fa1_spec(q.shape[-1])
    if backend in ("auto", "cuda"):
        return fa1_cuda(q, k, v, causal, softmax_scale, spec)
    if backend in ("triton", "torch"):
        raise NotImplementedError(
            f"backend={backend!r} is not shipped by the MI355X build: only the HIP path (backend='cuda'/'auto') exists"
        )
    raise ValueError(backend)
