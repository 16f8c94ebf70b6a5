"""CPU oracle for the FlashAttention forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package
(`flashattention-pytorch_amd/`) may import this module; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
there only as the checker / the timed CPU baseline.

Parity status: PINNED.  `tests/golden/*.npz` were produced in the build
container by importing the reference's own Python
(`/root/reference/src/common/correctness.py`, `src/fa1/torch/impl.py`) with
`tests/golden/make_golden.py`; `tests/test_oracle_golden.py` checks every
function below against those vectors.

What is restated (reference file:line, paths relative to /root/reference):

* `exact_attention`            <- `src/common/correctness.py:5-24`
      (`reference_attention`): fp32 scores = QK^T * scale, causal mask
      col > row -> -inf (`src/common/mask.py:6-12`), softmax, P@V cast back to
      the input dtype, lse = logsumexp (natural log, fp32).
      The reference's causal branch is broken for 3-D input (the
      mask helper reads shape[0], shape[1] of a (BH, N, N) tensor); this
      restatement applies the mask per (b,h) slice, which is what the helper
      does on the 2-D tiles it was written for.
* `exact_attention_backward`   <- `src/common/correctness.py:26-34`
      (`reference_backward`): autograd of the above with `o.backward(do)`.
      Written here in closed form (no autograd) in fp32 or fp64.
* `tiled_forward`              <- `src/fa1/torch/impl.py:26-68`
      (`fa1_forward_torch`): Q-tile outer / K-tile inner online softmax with
      un-normalised O until the epilogue.
* `tiled_backward`             <- `src/fa1/torch/impl.py:70-115`
      (`fa1_backward_torch`): dvec = rowsum(dO*O); K-tile outer / Q-tile inner
      recomputation from lse.
  The FA2 torch/csrc variants (`src/fa2/torch/impl.py:57,62,107-112`,
  `csrc/fa2/fa2_fwd.cu:92-99`, `csrc/fa?/fa?_bwd.cu:80`) carry the numeric
  defects listed in SURVEY §4.3 (D2-D4) and are NOT followed: FA1, FA2 and FA3
  are the same mathematical function and the correct statement is FA1's.
* `block_absmax_scale`, `quantize_e4m3_blockwise`
                               <- intent of `src/fa3/torch/impl.py:20-44`
      (per-block absmax scale, eps 1e-6, block = br for Q / bc for K).  The
      reference's "fp8" is an fp16 round trip; the oracle models a
      real OCP e4m3 quantisation, which is what the HIP fp8 path does.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

import torch

__all__ = [
    "exact_attention",
    "exact_attention_backward",
    "tiled_forward",
    "tiled_backward",
    "block_absmax_scale",
    "quantize_e4m3_blockwise",
    "fp8_attention",
    "fp8_attention_backward",
    "fp8_roundtrip",
    "attention_flops",
]


def _as3d(x: torch.Tensor) -> Tuple[torch.Tensor, Optional[Tuple[int, int]]]:
    # (B,H,N,d) -> (BH,N,d) view; reference: src/common/utils.py:3-7
    if x.dim() == 3:
        return x, None
    b, h, n, d = x.shape
    return x.reshape(b * h, n, d), (b, h)


def _causal_mask(n_q: int, n_k: int, device) -> torch.Tensor:
    r = torch.arange(n_q, device=device)[:, None]
    c = torch.arange(n_k, device=device)[None, :]
    return c > r  # True = masked; reference: src/common/mask.py:9-11


def exact_attention(q, k, v, causal=False, softmax_scale=None, math_dtype=torch.float32):
    """Dense attention. Returns (o in q.dtype, lse fp32). Shapes follow the input rank."""
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** -0.5
    qb, bh_shape = _as3d(q)
    kb, _ = _as3d(k)
    vb, _ = _as3d(v)
    n = qb.shape[1]
    s = torch.matmul(qb.to(math_dtype), kb.to(math_dtype).transpose(-2, -1)) * softmax_scale
    if causal:
        s = s.masked_fill(_causal_mask(n, kb.shape[1], s.device)[None], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    o = torch.matmul(p, vb.to(math_dtype)).to(q.dtype)
    lse = lse.to(torch.float32)
    if bh_shape is not None:
        b, h = bh_shape
        o = o.reshape(b, h, n, -1)
        lse = lse.reshape(b, h, n)
    return o, lse


def exact_attention_backward(q, k, v, do, causal=False, softmax_scale=None, math_dtype=torch.float32):
    """Closed-form gradient of `exact_attention`'s `o` w.r.t. q, k, v for upstream `do`.

    Matches what autograd gives for `o.backward(do)` in the reference
    (`src/common/correctness.py:26-34`): the output cast to q.dtype is treated
    as identity, the gradients are returned in the input dtype.
    Returns (dq, dk, dv, o, lse).
    """
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** -0.5
    qb, bh_shape = _as3d(q)
    kb, _ = _as3d(k)
    vb, _ = _as3d(v)
    dob, _ = _as3d(do)
    n = qb.shape[1]
    qf, kf, vf, dof = (t.to(math_dtype) for t in (qb, kb, vb, dob))
    s = torch.matmul(qf, kf.transpose(-2, -1)) * softmax_scale
    if causal:
        s = s.masked_fill(_causal_mask(n, kf.shape[1], s.device)[None], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    of = torch.matmul(p, vf)
    dv = torch.matmul(p.transpose(-2, -1), dof)
    dp = torch.matmul(dof, vf.transpose(-2, -1))
    delta = (dof * of).sum(dim=-1, keepdim=True)
    ds = p * (dp - delta)
    dq = torch.matmul(ds, kf) * softmax_scale
    dk = torch.matmul(ds.transpose(-2, -1), qf) * softmax_scale
    o = of.to(q.dtype)
    dq, dk, dv = dq.to(q.dtype), dk.to(k.dtype), dv.to(v.dtype)
    lse = lse.to(torch.float32)
    if bh_shape is not None:
        b, h = bh_shape
        dq, dk, dv, o = (t.reshape(b, h, n, -1) for t in (dq, dk, dv, o))
        lse = lse.reshape(b, h, n)
    return dq, dk, dv, o, lse


def tiled_forward(q, k, v, causal, softmax_scale, br, bc):
    """Online-softmax tile loop over (bh, Q tile, K tile); q,k,v are (BH,N,d).

    Follows `src/fa1/torch/impl.py:26-68`, with one deliberate difference: the
    causal mask is applied to every tile that intersects the diagonal, not only
    when the tile's first row lies inside the column tile (the reference's
    condition at :50 is correct only for br <= bc).
    """
    bh, n, d = q.shape
    o = torch.empty_like(q)
    lse = torch.empty((bh, n), dtype=torch.float32, device=q.device)
    for b in range(bh):
        for r0 in range(0, n, br):
            r1 = min(r0 + br, n)
            qi = q[b, r0:r1].float()
            m = torch.full((r1 - r0,), float("-inf"))
            l = torch.zeros(r1 - r0)
            acc = torch.zeros(r1 - r0, d)
            for c0 in range(0, n, bc):
                if causal and c0 >= r1:  # whole tile above the diagonal
                    break
                c1 = min(c0 + bc, n)
                s = (qi @ k[b, c0:c1].float().T) * softmax_scale
                if causal and c1 - 1 > r0:  # tile touches the diagonal
                    rr = torch.arange(r0, r1)[:, None]
                    cc = torch.arange(c0, c1)[None, :]
                    s = s.masked_fill(cc > rr, float("-inf"))
                m_new = torch.maximum(m, s.amax(dim=-1))
                p = torch.exp(s - m_new[:, None])
                alpha = torch.exp(m - m_new)
                l = alpha * l + p.sum(dim=-1)
                acc = alpha[:, None] * acc + p @ v[b, c0:c1].float()
                m = m_new
            o[b, r0:r1] = (acc / l[:, None]).to(q.dtype)
            lse[b, r0:r1] = m + torch.log(l)
    return o, lse


def tiled_backward(q, k, v, o, do, lse, causal, softmax_scale, br, bc):
    """Recomputation backward, K tile outer / Q tile inner (`src/fa1/torch/impl.py:70-115`)."""
    bh, n, d = q.shape
    dq = torch.zeros(bh, n, d)
    dk = torch.zeros(bh, n, d)
    dv = torch.zeros(bh, n, d)
    delta = (do.float() * o.float()).sum(dim=-1)
    for b in range(bh):
        for c0 in range(0, n, bc):
            c1 = min(c0 + bc, n)
            kj = k[b, c0:c1].float()
            vj = v[b, c0:c1].float()
            dkj = torch.zeros(c1 - c0, d)
            dvj = torch.zeros(c1 - c0, d)
            for r0 in range(0, n, br):
                r1 = min(r0 + br, n)
                if causal and c0 >= r1:  # every (row, col) of this pair is masked
                    continue
                qi = q[b, r0:r1].float()
                doi = do[b, r0:r1].float()
                s = (qi @ kj.T) * softmax_scale
                if causal and c1 - 1 > r0:
                    rr = torch.arange(r0, r1)[:, None]
                    cc = torch.arange(c0, c1)[None, :]
                    s = s.masked_fill(cc > rr, float("-inf"))
                p = torch.exp(s - lse[b, r0:r1, None])
                dvj += p.T @ doi
                dp = doi @ vj.T
                ds = p * (dp - delta[b, r0:r1, None])
                dq[b, r0:r1] += (ds @ kj) * softmax_scale
                dkj += (ds.T @ qi) * softmax_scale
            dk[b, c0:c1] = dkj
            dv[b, c0:c1] = dvj
    return dq.to(q.dtype), dk.to(k.dtype), dv.to(v.dtype)


# --------------------------------------------------------------------------- fp8 (FA3-style) model


def block_absmax_scale(x: torch.Tensor, block: int, eps: float = 1e-6) -> torch.Tensor:
    """Per (bh, row-block) absmax, clamped below by eps (`src/fa3/torch/impl.py:20-31`)."""
    bh, n, d = x.shape
    nb = (n + block - 1) // block
    out = torch.empty(bh, nb, dtype=torch.float32)
    for i in range(nb):
        blk = x[:, i * block : min((i + 1) * block, n)].float()
        out[:, i] = blk.abs().amax(dim=(1, 2)).clamp_min(eps)
    return out


E4M3_MAX = 448.0


def quantize_e4m3_blockwise(x: torch.Tensor, block: int):
    """Quantise to OCP e4m3 with one scale per (bh, row block): x ~= xq * scale.

    scale = absmax / 448 so the block's largest element maps to the e4m3
    maximum.  Returns (xq as float32 holding exactly-representable e4m3 values,
    scale (bh, nb) fp32).
    """
    bh, n, d = x.shape
    amax = block_absmax_scale(x, block)
    scale = amax / E4M3_MAX
    xq = torch.empty(bh, n, d, dtype=torch.float32)
    for i in range(amax.shape[1]):
        sl = slice(i * block, min((i + 1) * block, n))
        y = x[:, sl].float() / scale[:, i, None, None]
        xq[:, sl] = y.to(torch.float8_e4m3fn).float()
    return xq, scale


def incoherent_signs(d: int) -> torch.Tensor:
    """The fixed +-1 vector of the rotation below: sign(e) = bit (0x9E3779B1 * e mod 2^32) >> 27 of 0x5A3C96E1."""
    e = np.arange(d, dtype=np.uint64)
    h = ((e * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(27)
    bit = (np.uint64(0x5A3C96E1) >> h) & np.uint64(1)
    return torch.from_numpy(1.0 - 2.0 * bit.astype(np.float64)).float()


def incoherent_rotate(x: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """x -> x . diag(s) . H / sqrt(d): the incoherent processing the reference's FA3 path intends before quantising Q
    and K (`src/fa3/torch/impl.py:20-45`: random signs, then a Hadamard transform; its own transform is not orthogonal,
    SURVEY D6).  Orthogonal, so (Q R)(K R)^T = Q K^T; it spreads an outlier channel over all d channels, which is what
    makes one e4m3 scale per row block adequate.  H = Sylvester Hadamard matrix (d a power of two), fp32 arithmetic."""
    d = x.shape[-1]
    assert d & (d - 1) == 0, "Hadamard rotation needs a power-of-two head dim"
    h = torch.ones(1, 1)
    while h.shape[0] < d:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    s = incoherent_signs(d)
    xf = x.float()
    if inverse:
        return (xf @ h) * s / math.sqrt(d)
    return ((xf * s) @ h) / math.sqrt(d)


def quantize_e4m3_blockwise_pow2(x: torch.Tensor, block: int):
    """As `quantize_e4m3_blockwise` with the scale rounded UP to a power of two (2^e >= absmax / 448): the form the block-scaled
    MFMA applies by itself (E8M0 scale operand), used for V by the all-e4m3 kernel (csrc/fa_fwd_fp8.hip: fp8_quant_v_kernel)."""
    amax = block_absmax_scale(x, block)
    scale = torch.exp2(torch.ceil(torch.log2(amax.double() / E4M3_MAX))).float()
    xq = torch.empty(x.shape, dtype=torch.float32)
    for i in range(amax.shape[1]):
        sl = slice(i * block, min((i + 1) * block, x.shape[1]))
        xq[:, sl] = (x[:, sl].float() / scale[:, i, None, None]).to(torch.float8_e4m3fn).float()
    return xq, scale


def fp8_roundtrip(q, k, v, block_q, block_k, rotate: bool = True, quantize_v: bool = True, v_pow2: bool = False):
    """(Q~, K~, V~): what the e4m3 path computes attention OF — each tensor quantised to e4m3 with one absmax scale per row
    block and dequantised (fp32 results).  Q and K are rotated by `incoherent_rotate` around the quantisation when the head dim
    is a power of two (the reference skips its rotation otherwise too, src/fa3/torch/impl.py:60-61) and rotated back, which
    leaves Q~ K~^T what the rotated tensors give: (Q^ R^T)(K^ R^T)^T = Q^ K^^T.  V is quantised as it is
    (src/fa3/torch/impl.py:125-131: sv / vb2).  rotate=False: the library's option fp8_rot = 2."""
    d = q.shape[-1]
    rot = rotate and (d & (d - 1)) == 0

    def rt(x, block, r, pow2=False):
        xr = incoherent_rotate(x) if r else x.float()
        xq, sc = (quantize_e4m3_blockwise_pow2 if pow2 else quantize_e4m3_blockwise)(xr, block)
        y = xq * torch.repeat_interleave(sc, block, dim=1)[:, : x.shape[1], None]
        return incoherent_rotate(y, inverse=True) if r else y

    return rt(q, block_q, rot), rt(k, block_k, rot), (rt(v, block_k, False, v_pow2) if quantize_v else v.float())


def fp8_attention(q, k, v, causal, softmax_scale, block_q, block_k, rotate: bool = True, quantize_v: bool = True,
                  p_e4m3: bool = False):
    """Attention with Q, K and V quantised to e4m3 per row block; the numerical model of the HIP `fa3_forward(fp8=True)` path.
    p_e4m3=False: P and the accumulation in fp32 — the path whose P.V product is 16-bit (every head dim but 128, and option
    fp8_pv = 1 there).  p_e4m3=True: the all-e4m3 kernel (d = 128 default): V with a power-of-two block scale and the
    probabilities p = exp(s - rowmax) rounded to e4m3 in the P.V product, the row sum (and lse) from the unrounded p.  (The
    kernel rounds p relative to a running maximum that may lag the true one by up to 2^8, so its roundings differ from these
    element by element; the model has the same error statistics, not the same bits.)
    This is this repo's model of the reference's INTENT (its own fp8 branch models no 8-bit rounding and its rotation is not
    orthogonal: SURVEY D6, D7) — fp8 parity is not pinned by a reference fixture."""
    qd, kd, vd = fp8_roundtrip(q, k, v, block_q, block_k, rotate, quantize_v, v_pow2=p_e4m3)
    if not p_e4m3:
        o, lse = exact_attention(qd, kd, vd, causal, softmax_scale)
        return o.to(v.dtype), lse
    s = torch.matmul(qd.double(), kd.double().transpose(-2, -1)) * softmax_scale
    if causal:
        s = s.masked_fill(_causal_mask(s.shape[-2], s.shape[-1], s.device)[None], float("-inf"))
    m = s.amax(dim=-1, keepdim=True)
    p = torch.exp(s - m)
    p8 = p.float().to(torch.float8_e4m3fn).double()
    o = torch.matmul(p8, vd.double()) / p.sum(dim=-1, keepdim=True)
    lse = (m.squeeze(-1) + torch.log(p.sum(dim=-1))).float()
    return o.float().to(v.dtype), lse


def fp8_attention_backward(q, k, v, dout, causal, softmax_scale, block_q, block_k, rotate: bool = True, quantize_v: bool = True):
    """(dq, dk, dv, o, lse) of the function `fp8_attention` evaluates, taken at the round-tripped tensors and handed to q, k, v
    themselves (straight-through over the rounding), as the reference's fa3_backward does (csrc/fa3/fa3_bwd.cu:134-146)."""
    qd, kd, vd = fp8_roundtrip(q, k, v, block_q, block_k, rotate, quantize_v)
    return exact_attention_backward(qd, kd, vd, dout, causal, softmax_scale, math_dtype=torch.float64)


# --------------------------------------------------------------------------- FLOP accounting


def attention_flops(bh: int, n: int, d: int, direction: str, causal: bool = False, convention: str = "algorithmic"):
    """FLOPs of one pass.

    convention="algorithmic": fwd 4*N^2*d, bwd 10*N^2*d (5 GEMMs), fwd+bwd 14*N^2*d,
    causal scaled by (N+1)/(2N)  (SURVEY §8d).
    convention="reference": `benchmarks/bench_utils.py:210-215` — forward 4*B*H*N^2*d,
    "backward" (= fwd+bwd timed together) 8*B*H*N^2*d, no causal discount.
    """
    if convention == "reference":
        factor = 4.0 if direction == "forward" else 8.0
        return factor * bh * n * n * d
    factor = {"forward": 4.0, "backward": 10.0, "fwd+bwd": 14.0}[direction]
    f = factor * bh * n * n * d
    if causal:
        f *= (n + 1) / (2.0 * n)
    return f


# ---- extended attention: Nq != Nk causal offset, dense mask, block-sparse mask, dropout ----
# Restates the attention core of the reference's notebook model: scores = tau * q k^T / sqrt(d), masked_fill(mask == 0,
# -inf) (src/fa3/torch/flashattention_pytorch.py:134-141, 80-84), look_ahead_mask_ (:176-190), tiles with
# block_sparse_mask[i, j] == 0 skipped (:123-125), dropout of the probabilities: keep where rnd > p, scale 1 / (1 - p)
# (src/common/dropout.py:9-15, flashattention_pytorch.py:85-87 — the dense branch's standard dropout, O = dropout(softmax) V;
# the tiled branch's renormalisation by the kept sum, :155-163, is a defect and is not reproduced).

def dropout_keep(bh, nq, nk, p, seed):
    """(bh, nq, nk) boolean keep mask of the HIP kernels' counter-based generator (csrc/fa_ex_common.h: ex_keep): one
    splitmix64 value per 2 x 2 quad of (row, key) elements — counter (bh * ceil(nq/2) + row//2) << 32 | key//2, plus
    seed * G + G — and 16 uniform bits per element (field 2 * (row & 1) + (key & 1)); keep iff u >= floor(65536 p) + 1.
    numpy uint64 arithmetic wraps like the device's."""
    import numpy as np

    if p <= 0.0:
        return torch.ones((bh, nq, nk), dtype=torch.bool)
    thr = int(p * 65536.0) + 1
    with np.errstate(over="ignore"):
        b = np.arange(bh, dtype=np.uint64)[:, None, None]
        r = np.arange(nq, dtype=np.uint64)[None, :, None]
        c = np.arange(nk, dtype=np.uint64)[None, None, :]
        hi = (b * np.uint64((nq + 1) // 2) + (r >> np.uint64(1))) & np.uint64(0xFFFFFFFF)
        g = np.uint64(0x9E3779B97F4A7C15)
        z = ((hi << np.uint64(32)) | (c >> np.uint64(1))) + np.uint64(seed % (1 << 64)) * g + g
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        field = np.uint64(2) * (r & np.uint64(1)) + (c & np.uint64(1))
        u = (z >> (np.uint64(16) * field)) & np.uint64(0xFFFF)
    return torch.from_numpy(u >= np.uint64(thr))


def extended_visible(bh, nq, nk, causal=False, mask=None, block_mask=None, br=128, bc=128):
    """(bh, nq, nk) boolean: which (query, key) pairs take part."""
    vis = torch.ones((bh, nq, nk), dtype=torch.bool)
    if causal:
        qi = torch.arange(nq).unsqueeze(1)
        kj = torch.arange(nk).unsqueeze(0)
        vis &= (kj <= qi + (nk - nq)).unsqueeze(0)
    if mask is not None:
        m = (mask != 0)
        vis &= m.unsqueeze(0) if m.dim() == 2 else m
    if block_mask is not None:
        bm = (block_mask != 0)
        full = bm.repeat_interleave(br, dim=0)[:nq].repeat_interleave(bc, dim=1)[:, :nk]
        vis &= full.unsqueeze(0)
    return vis


def extended_attention(q, k, v, causal=False, softmax_scale=None, mask=None, block_mask=None, br=128, bc=128, dropout_p=0.0,
                       seed=0, math_dtype=torch.float64):
    """(o, lse) in `math_dtype`; rows without a visible key: o = 0, lse = -inf (the kernels' convention)."""
    bh, nq, d = q.shape
    nk = k.shape[1]
    scale = d ** -0.5 if softmax_scale is None else softmax_scale
    qf, kf, vf = q.to(math_dtype), k.to(math_dtype), v.to(math_dtype)
    s = qf @ kf.transpose(1, 2) * scale
    vis = extended_visible(bh, nq, nk, causal, mask, block_mask, br, bc)
    s = s.masked_fill(~vis, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    dead = ~vis.any(dim=-1)
    p = torch.exp(s - lse.masked_fill(dead, 0.0).unsqueeze(-1))
    p = p.masked_fill(~vis, 0.0)
    if dropout_p > 0.0:
        p = p * dropout_keep(bh, nq, nk, dropout_p, seed).to(math_dtype) / (1.0 - dropout_p)
    o = p @ vf
    return o, lse


def extended_attention_backward(q, k, v, do, **kw):
    """(dq, dk, dv, o, lse) by autograd through extended_attention (fp64), results cast to q.dtype except lse."""
    math_dtype = kw.get("math_dtype", torch.float64)
    qf, kf, vf = (t.detach().to(math_dtype).requires_grad_(True) for t in (q, k, v))
    o, lse = extended_attention(qf, kf, vf, **kw)
    (o * do.to(math_dtype)).sum().backward()
    return qf.grad.to(q.dtype), kf.grad.to(q.dtype), vf.grad.to(q.dtype), o.detach().to(q.dtype), lse.detach().float()
