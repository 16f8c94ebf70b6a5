"""GPU parity of the extended attention path through the C-ABI (`fa_ex_forward` / `fa_ex_backward`):
against the golden vectors of the reference's notebook code, and against the fp64 oracle — forward and backward — for
Nq != Nk with the bottom-right causal mask, dense masks (shared and per (b,h)), block-sparse masks, dropout (the kernels'
counter-based mask is reproduced bit for bit by the oracle, so dropout is an exact-parity case, not a statistical one) and
all of them together; ragged sizes, head dims 40 / 64 / 128, three dtypes, rows without any visible key."""
import pytest
import torch

from oracle import attention_oracle as orc
from tests.helpers import dtype_tolerances, ex_golden_tags, load_ex_golden, max_abs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ex_golden_tags())
def test_hip_matches_reference_notebook_vectors(tag, device):
    from common.attention_ex import flash_attention_ex

    meta, g = load_ex_golden(tag)
    q, k, v = (g[x].to(device) for x in "qkv")              # (B, H, N, d), as the model passes them
    mask = g["mask"].to(device) if "mask" in g else None
    o = flash_attention_ex(q, k, v, tau=meta["tau"], mask=mask, block_sparse_mask=g["block_mask"].to(device),
                           block_size=meta["block_size"])
    assert o.shape == q.shape
    assert max_abs(o.cpu(), g["o"]) < 1e-4                   # fp32 bar of the reference's tests (tests/utils.py:36)
    if meta["causal"] and not meta["dense_mask"]:            # the same mask, not materialised
        o2 = flash_attention_ex(q, k, v, tau=meta["tau"], causal=True, block_sparse_mask=g["block_mask"].to(device),
                                block_size=meta["block_size"])
        assert max_abs(o2.cpu(), g["o"]) < 1e-4


def _case(bh, nq, nk, d, dtype, seed, mask_kind=None, block=None, density=0.6):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn((bh, nq, d), generator=g).to(dtype)
    k = torch.randn((bh, nk, d), generator=g).to(dtype)
    v = torch.randn((bh, nk, d), generator=g).to(dtype)
    do = torch.randn((bh, nq, d), generator=g).to(dtype)
    mask = bmask = None
    if mask_kind == "shared":
        mask = (torch.rand((nq, nk), generator=g) < density).to(torch.uint8)
    elif mask_kind == "per_bh":
        mask = (torch.rand((bh, nq, nk), generator=g) < density).to(torch.uint8)
    if block is not None:
        br, bc = block
        bmask = (torch.rand(((nq + br - 1) // br, (nk + bc - 1) // bc), generator=g) < density).to(torch.uint8)
    return q, k, v, do, mask, bmask


CASES = [
    # bh, nq, nk, d, causal, mask_kind, block (br, bc), dropout_p
    (2, 40, 72, 32, True, None, None, 0.0),          # keys longer than queries: cached past keys
    (2, 100, 37, 64, True, None, None, 0.0),         # queries longer than keys: the first rows see nothing
    (3, 65, 130, 40, False, "shared", None, 0.0),
    (2, 70, 70, 64, False, "per_bh", None, 0.0),
    (2, 200, 300, 64, False, None, (32, 64), 0.0),
    (1, 257, 257, 128, True, None, (64, 64), 0.0),
    (2, 90, 120, 64, False, None, None, 0.3),
    (2, 128, 128, 128, True, None, None, 0.1),
    (2, 150, 210, 48, True, "shared", (32, 32), 0.25),   # everything at once
    (1, 1, 5, 16, False, None, None, 0.0),
    (1, 33, 1, 8, True, None, None, 0.0),
]


@pytest.mark.parametrize("bh,nq,nk,d,causal,mask_kind,block,p", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_forward_and_backward_match_the_oracle(bh, nq, nk, d, causal, mask_kind, block, p, dtype, device):
    import flashattention_lab_cuda as ext

    q, k, v, do, mask, bmask = _case(bh, nq, nk, d, dtype, seed=1000 + nq + nk + d, mask_kind=mask_kind, block=block)
    br, bc = block if block is not None else (128, 128)
    scale, seed = 0.9 * d ** -0.5, 4242 + nq
    kw = dict(causal=causal, softmax_scale=scale, mask=mask, block_mask=bmask, br=br, bc=bc, dropout_p=p, seed=seed)
    rq, rk, rv, ro, rlse = orc.extended_attention_backward(q, k, v, do, **kw)
    dev = lambda t: None if t is None else t.to(device)
    o, lse = ext.ex_forward(dev(q), dev(k), dev(v), causal, scale, dev(mask), dev(bmask), br, bc, p, seed)
    dq, dk, dv = ext.ex_backward(dev(q), dev(k), dev(v), o, dev(do), lse, causal, scale, dev(mask), dev(bmask), br, bc, p, seed)
    tol = dtype_tolerances(dtype)
    torch.testing.assert_close(o.cpu(), ro, **tol)
    live = torch.isfinite(rlse)
    assert torch.equal(torch.isfinite(lse.cpu()), live)          # rows without a visible key: lse = -inf, exactly those
    torch.testing.assert_close(lse.cpu()[live], rlse[live], rtol=1e-3, atol=1e-3)
    assert torch.equal(o.cpu()[~live], torch.zeros_like(o.cpu()[~live]))
    for name, a, b in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        assert torch.isfinite(a.float()).all(), name
        torch.testing.assert_close(a.cpu(), b, **tol)
    if dtype == torch.float32:
        assert max_abs(o.cpu(), ro) < 5e-5 and max(max_abs(dq.cpu(), rq), max_abs(dk.cpu(), rk), max_abs(dv.cpu(), rv)) < 2e-4


BIG_CASES = [
    # several 256-row / 128-key tiles per (b,h): tile skipping, the diagonal across tiles, both mask fetch paths
    (2, 700, 900, 64, True, None, None, 0.0, 0.6),
    (2, 900, 700, 128, True, None, None, 0.0, 0.6),          # Nq > Nk: the first 200 rows see nothing
    (1, 600, 1000, 128, False, "shared", None, 0.0, 0.5),    # Nk % 4 == 0: dword mask loads
    (2, 513, 777, 64, False, "per_bh", None, 0.0, 0.5),      # byte mask loads
    (2, 600, 1040, 128, False, "per_bh", None, 0.0, 0.5),    # Nk % 16 == 0: one 16-byte mask load per lane and block (half-wave swap / LDS image)
    (2, 513, 784, 64, True, "per_bh", None, 0.0, 0.7),       # the same at d = 64, under the causal flag too
    (1, 1024, 1024, 128, True, None, (128, 128), 0.0, 0.3),  # most tiles dead
    (1, 800, 1100, 64, False, None, (64, 32), 0.0, 0.2),
    (2, 520, 640, 128, True, "shared", (32, 32), 0.2, 0.7),  # everything at once
    (2, 384, 512, 64, False, None, None, 0.5, 0.6),
    (2, 300, 300, 72, True, None, (96, 160), 0.1, 0.5),      # blocks that do not divide the tiles, padded head dim
    (2, 640, 640, 128, True, None, None, 0.0, 0.5),          # square, no extras: "mfma" hands this one to the plain kernels
    (3, 500, 500, 64, False, None, None, 0.0, 0.5),
]


@pytest.mark.parametrize("bh,nq,nk,d,causal,mask_kind,block,p,density", BIG_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("path", ["mfma", "mfma_only", "exact"])
def test_multi_tile_cases_on_both_kernel_families(bh, nq, nk, d, causal, mask_kind, block, p, density, dtype, path, device):
    """16-bit tensors take the MFMA kernels (csrc/fa_ex_mfma.hip) by default; option ex_path pins either family.  Both must
    match the fp64 oracle, dropout masks included, and mark exactly the rows without a visible key."""
    import flashattention_lab_cuda as ext

    q, k, v, do, mask, bmask = _case(bh, nq, nk, d, dtype, seed=77 + nq + nk + d, mask_kind=mask_kind, block=block, density=density)
    br, bc = block if block is not None else (128, 128)
    scale, seed = d ** -0.5, 99 + nk
    kw = dict(causal=causal, softmax_scale=scale, mask=mask, block_mask=bmask, br=br, bc=bc, dropout_p=p, seed=seed)
    rq, rk, rv, ro, rlse = orc.extended_attention_backward(q, k, v, do, **kw)
    dev = lambda t: None if t is None else t.to(device)
    ext.set_option("ex_path", {"mfma": 2, "mfma_only": 3, "exact": 1}[path])   # 3: never the plain kernels
    try:
        o, lse = ext.ex_forward(dev(q), dev(k), dev(v), causal, scale, dev(mask), dev(bmask), br, bc, p, seed)
        dq, dk, dv = ext.ex_backward(dev(q), dev(k), dev(v), o, dev(do), lse, causal, scale, dev(mask), dev(bmask), br, bc, p, seed)
    finally:
        ext.set_option("ex_path", 0)
    tol = dtype_tolerances(dtype)
    torch.testing.assert_close(o.cpu(), ro, **tol)
    live = torch.isfinite(rlse)
    assert torch.equal(torch.isfinite(lse.cpu()), live)
    torch.testing.assert_close(lse.cpu()[live], rlse[live], rtol=1e-3, atol=1e-3)
    assert torch.equal(o.cpu()[~live], torch.zeros_like(o.cpu()[~live]))
    for name, a, b in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        assert torch.isfinite(a.float()).all(), name
        torch.testing.assert_close(a.cpu(), b, **tol)
    assert torch.equal(dq.cpu()[~live], torch.zeros_like(dq.cpu()[~live]))


@pytest.mark.parametrize("nq,nk", [(700, 900), (512, 512), (900, 640)])
def test_structured_dense_mask_takes_the_block_shortcuts_and_matches_the_causal_flag(nq, nk, device):
    """look_ahead_mask handed over as bytes (what the notebook model does): most 32 x 32 blocks are all-visible or
    all-masked and take the kernels' wave-uniform shortcuts (no per-byte work, no products for a block nobody sees); the
    result must be the causal flag's within rounding."""
    import flashattention_lab_cuda as ext
    from common.attention_ex import look_ahead_mask

    q, k, v, do, _, _ = _case(2, nq, nk, 128, torch.bfloat16, seed=5 + nq)
    qd, kd, vd, dod = (t.to(device) for t in (q, k, v, do))
    mask = look_ahead_mask(nq, nk, device=device)[0, 0].to(torch.uint8).contiguous()
    ext.set_option("ex_path", 3)
    try:
        o1, l1 = ext.ex_forward(qd, kd, vd, True, 0.09)
        o2, l2 = ext.ex_forward(qd, kd, vd, False, 0.09, mask=mask)
        g1 = ext.ex_backward(qd, kd, vd, o1, dod, l1, True, 0.09)
        g2 = ext.ex_backward(qd, kd, vd, o2, dod, l2, False, 0.09, mask=mask)
    finally:
        ext.set_option("ex_path", 0)
    # the same products in the same order; with ragged query tiles the lazy rescale of the online softmax fires on other
    # tiles (rows past Nq are masked by the dense mask, visible to the flag's arithmetic), so the last bit may differ
    assert max_abs(o1, o2) < 4e-3 and max_abs(l1[torch.isfinite(l1)], l2[torch.isfinite(l2)]) < 1e-4
    assert torch.equal(torch.isfinite(l1), torch.isfinite(l2))
    for a, b in zip(g1, g2):
        assert max_abs(a, b) < 2e-2 * max(1.0, float(a.float().abs().max()))


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("nq,nk", [(1024, 1536), (1000, 1300), (1536, 1024)])
def test_big_launches_without_extras_take_the_plain_kernels_with_separate_row_counts(nq, nk, causal, device):
    """Nq != Nk, no mask, no dropout, d = 128, a launch big enough for the 256-row tiles: the staggered forward and the stream
    backward of the plain path run with nk keys (causal only with Nk >= Nq; otherwise the extended kernels).  Checked against
    the extended MFMA kernels on every unit and against the fp64 oracle on two."""
    import flashattention_lab_cuda as ext

    bh, d = 64, 128
    g = torch.Generator().manual_seed(nq + nk)
    q = torch.randn((bh, nq, d), generator=g).to(torch.bfloat16)
    k = torch.randn((bh, nk, d), generator=g).to(torch.bfloat16)
    v = torch.randn((bh, nk, d), generator=g).to(torch.bfloat16)
    do = torch.randn((bh, nq, d), generator=g).to(torch.bfloat16)
    qd, kd, vd, dod = (t.to(device) for t in (q, k, v, do))
    scale = d ** -0.5
    o, lse = ext.ex_forward(qd, kd, vd, causal, scale)
    grads = ext.ex_backward(qd, kd, vd, o, dod, lse, causal, scale)
    ext.set_option("ex_path", 3)
    try:
        o3, lse3 = ext.ex_forward(qd, kd, vd, causal, scale)
        grads3 = ext.ex_backward(qd, kd, vd, o3, dod, lse3, causal, scale)
    finally:
        ext.set_option("ex_path", 0)
    live = torch.isfinite(lse3)
    assert torch.equal(torch.isfinite(lse), live)
    assert max_abs(o, o3) < 8e-3 and max_abs(lse[live], lse3[live]) < 1e-3
    for a, b in zip(grads, grads3):
        assert torch.isfinite(a.float()).all() and max_abs(a, b) < 3e-2 * max(1.0, float(b.float().abs().max()))
    rq, rk, rv, ro, rlse = orc.extended_attention_backward(q[:2], k[:2], v[:2], do[:2], causal=causal, softmax_scale=scale)
    tol = dtype_tolerances(torch.bfloat16)
    torch.testing.assert_close(o[:2].cpu(), ro, **tol)
    lv = torch.isfinite(rlse)
    torch.testing.assert_close(lse[:2].cpu()[lv], rlse[lv], rtol=1e-3, atol=1e-3)
    for a, b in zip(grads, (rq, rk, rv)):
        torch.testing.assert_close(a[:2].cpu(), b, **tol)


def test_mfma_family_refuses_what_it_does_not_cover(device):
    import flashattention_lab_cuda as ext

    q = torch.randn((1, 64, 64), device=device)
    ext.set_option("ex_path", 2)
    try:
        with pytest.raises(RuntimeError):
            ext.ex_forward(q, q, q, False, 0.125)                                   # fp32 tensors
        with pytest.raises(RuntimeError):
            h = q.to(torch.bfloat16)
            ext.ex_forward(h, h, h, False, 0.125, block_mask=torch.ones((2, 2), dtype=torch.uint8, device=device), br=48, bc=48)
    finally:
        ext.set_option("ex_path", 0)


def test_extras_off_is_the_plain_path_and_dropout_is_reproducible(device):
    """No extras: the extended entry point computes what the plain one does.  Dropout: the same seed gives the same result
    bit for bit (forward and backward), another seed another mask, and E[o] over seeds approaches the undropped o."""
    import flashattention_lab_cuda as ext

    q, k, v, do, _, _ = _case(3, 96, 96, 64, torch.float32, seed=5)
    qd, kd, vd, dod = (t.to(device) for t in (q, k, v, do))
    o, lse = ext.ex_forward(qd, kd, vd, True, 0.125)
    o0, lse0 = ext.forward(qd, kd, vd, True, 0.125, 128, 128)
    assert max_abs(o, o0) < 2e-5 and max_abs(lse, lse0) < 2e-5
    a = ext.ex_forward(qd, kd, vd, False, 0.125, dropout_p=0.5, seed=11)
    b = ext.ex_forward(qd, kd, vd, False, 0.125, dropout_p=0.5, seed=11)
    c = ext.ex_forward(qd, kd, vd, False, 0.125, dropout_p=0.5, seed=12)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and not torch.equal(a[0], c[0])
    assert torch.equal(a[1], c[1])                                # lse does not see the dropout
    ga = ext.ex_backward(qd, kd, vd, a[0], dod, a[1], False, 0.125, dropout_p=0.5, seed=11)
    gb = ext.ex_backward(qd, kd, vd, b[0], dod, b[1], False, 0.125, dropout_p=0.5, seed=11)
    assert all(torch.equal(x, y) for x, y in zip(ga, gb))
    plain = ext.ex_forward(qd, kd, vd, False, 0.125)[0]
    mean = sum(ext.ex_forward(qd, kd, vd, False, 0.125, dropout_p=0.3, seed=s)[0] for s in range(200)) / 200
    assert max_abs(mean, plain) < 0.2


def test_autograd_wrapper_and_error_behaviour(device):
    from common.attention_ex import flash_attention_ex, look_ahead_mask

    g = torch.Generator().manual_seed(3)
    q = torch.randn((2, 3, 20, 32), generator=g).to(device).requires_grad_(True)
    k = torch.randn((2, 3, 50, 32), generator=g).to(device).requires_grad_(True)
    v = torch.randn((2, 3, 50, 32), generator=g).to(device).requires_grad_(True)
    o = flash_attention_ex(q, k, v, mask=look_ahead_mask(20, 50, device=device))
    o2 = flash_attention_ex(q, k, v, causal=True)
    assert o.shape == q.shape and torch.equal(o, o2)
    o.sum().backward()
    ro, _ = orc.extended_attention(q.detach().cpu().reshape(6, 20, 32), k.detach().cpu().reshape(6, 50, 32),
                                   v.detach().cpu().reshape(6, 50, 32), causal=True)
    assert max_abs(o.detach().cpu().reshape(6, 20, 32), ro) < 1e-4
    assert q.grad is not None and torch.isfinite(q.grad).all() and torch.isfinite(k.grad).all()
    import flashattention_lab_cuda as ext
    with pytest.raises(RuntimeError):   # k and v must agree, q and k must share BH and d
        ext.ex_forward(q.detach().reshape(6, 20, 32), k.detach().reshape(6, 50, 32), v.detach().reshape(6, 50, 32)[:, :40], False, 0.1)
    with pytest.raises(RuntimeError, match="mask"):
        ext.ex_forward(q.detach().reshape(6, 20, 32), k.detach().reshape(6, 50, 32), v.detach().reshape(6, 50, 32), False, 0.1,
                       mask=torch.ones(20, 49, dtype=torch.uint8, device=device))
    with pytest.raises(RuntimeError, match="dropout_p"):
        ext.ex_forward(q.detach().reshape(6, 20, 32), k.detach().reshape(6, 50, 32), v.detach().reshape(6, 50, 32), False, 0.1, dropout_p=1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_an_empty_side_gives_zero_gradients_and_minus_infinity_lse(dtype, device):
    """nq == 0 with nk > 0 (dK = dV = 0) and nk == 0 with nq > 0 (o = 0, lse = -inf, dQ = 0) — sums over nothing,
    not a launch with an empty grid; a block-sparse mask with an empty side does not divide by Br = 0 either."""
    import flashattention_lab_cuda as ext

    bh, n, d = 3, 40, 32
    g = torch.Generator().manual_seed(5)
    full = torch.randn((bh, n, d), generator=g).to(dtype).to(device)
    empty = torch.empty((bh, 0, d), dtype=dtype, device=device)
    # no queries
    o, lse = ext.ex_forward(empty, full, full, False, 0.2)
    assert o.shape == (bh, 0, d) and lse.shape == (bh, 0)
    dq, dk, dv = ext.ex_backward(empty, full, full, o, empty, lse, False, 0.2)
    assert dq.shape == (bh, 0, d) and not dk.any() and not dv.any()
    dq, dk, dv = ext.ex_backward(empty, full, full, o, empty, lse, True, 0.2, block_mask=torch.ones((0, 2), dtype=torch.uint8, device=device),
                                 br=0, bc=32, dropout_p=0.1, seed=3)
    assert not dk.any() and not dv.any()
    # no keys
    o, lse = ext.ex_forward(full, empty, empty, True, 0.2)
    assert not o.any() and torch.isinf(lse).all() and (lse < 0).all()
    dq, dk, dv = ext.ex_backward(full, empty, empty, o, full, lse, True, 0.2)
    assert not dq.any() and dk.shape == (bh, 0, d) and dv.shape == (bh, 0, d)
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape", ["key_padding_b", "key_padding_shared", "per_batch", "per_head_3d"])
def test_broadcast_masks_through_the_wrapper(shape, device):
    """masks that are only BROADCASTABLE to (B, H, Nq, Nk) — a key-padding mask (B, 1, 1, Nk) or (1, 1, 1, Nk), a per-batch
    mask (B, 1, Nq, Nk), a 3-D (H, Nq, Nk) one — give what ordinary broadcasting gives (the reference indexes the mask as 4-D and
    lets masked_fill broadcast, flashattention_pytorch.py:139-141)."""
    from common.attention_ex import flash_attention_ex

    b, h, nq, nk, d = 2, 3, 70, 90, 64
    g = torch.Generator().manual_seed(17)
    q = torch.randn((b, h, nq, d), generator=g).to(torch.bfloat16)
    k = torch.randn((b, h, nk, d), generator=g).to(torch.bfloat16)
    v = torch.randn((b, h, nk, d), generator=g).to(torch.bfloat16)
    if shape == "key_padding_b":
        mask = torch.ones((b, 1, 1, nk), dtype=torch.bool)
        mask[0, ..., 60:] = False
        mask[1, ..., 33:] = False
    elif shape == "key_padding_shared":
        mask = torch.ones((1, 1, 1, nk), dtype=torch.bool)
        mask[..., 77:] = False
    elif shape == "per_batch":
        mask = torch.rand((b, 1, nq, nk), generator=g) > 0.3
        mask[..., 0] = True
    else:
        mask = torch.rand((h, nq, nk), generator=g) > 0.3
        mask[..., 0] = True
    full = torch.broadcast_to(mask, (b, h, nq, nk)).reshape(b * h, nq, nk)
    o = flash_attention_ex(q.to(device), k.to(device), v.to(device), mask=mask.to(device))
    ro, _ = orc.extended_attention(q.reshape(b * h, nq, d), k.reshape(b * h, nk, d), v.reshape(b * h, nk, d), mask=full)
    assert o.shape == q.shape
    torch.testing.assert_close(o.cpu().float().reshape(b * h, nq, d), ro.float(), rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize("nq,nk,causal", [(1024, 1536, False), (1100, 1300, False), (1536, 1024, False), (1024, 1024, False),
                                          (1024, 1536, True), (1100, 1300, True), (1024, 1024, True)])
def test_plain_calls_hand_ds_over_on_the_extended_path_too(nq, nk, causal, device):
    """`fa_ex_backward` without mask and dropout on a launch big enough for it (> 256 row tiles, no causal mask)
    takes the dS hand-over — preparation launch, dK/dV kernel with dS stores, dQ product — also with Nq != Nk (round 2: those
    calls recomputed; round 3: under the shifted causal diagonal too, Nk >= Nq): the library profile shows the preparation launch, the workspace is fa_ex_backward_workspace_bytes_fast's,
    the gradients agree with the recomputing pass (option dq = 5) and with the fp64 oracle."""
    import flashattention_lab_cuda as ext

    bh, d = 72, 128
    g = torch.Generator().manual_seed(3 * nq + nk)
    q = torch.randn((bh, nq, d), generator=g).to(torch.bfloat16)
    k = torch.randn((bh, nk, d), generator=g).to(torch.bfloat16)
    v = torch.randn((bh, nk, d), generator=g).to(torch.bfloat16)
    do = torch.randn((bh, nq, d), generator=g).to(torch.bfloat16)
    qd, kd, vd, dod = (t.to(device) for t in (q, k, v, do))
    scale = d ** -0.5
    lib = ext._lib
    small = int(lib.fa_ex_backward_workspace_bytes(bh, nq, nk, d, 2))
    fast = int(lib.fa_ex_backward_workspace_bytes_fast(bh, nq, nk, d, 2, int(causal), 0))
    tiles = ((nq + 31) // 32) * (8 * ((nk + 255) // 256))
    assert fast == small + bh * tiles * 2048                                                    # the whole launch in one chunk here
    assert int(lib.fa_ex_backward_workspace_bytes_fast(bh, nq, nk, d, 2, int(causal), 1)) == small        # a mask / dropout: no hand-over
    if nq > nk:
        assert int(lib.fa_ex_backward_workspace_bytes_fast(bh, nq, nk, d, 2, 1, 0)) == small    # causal with Nk < Nq: not on these kernels
    o, lse = ext.ex_forward(qd, kd, vd, causal, scale)
    ext.release_workspace()
    ext.profile_enable(True)
    grads = ext.ex_backward(qd, kd, vd, o, dod, lse, causal, scale)
    torch.cuda.synchronize()
    prof = ext.profile_report()
    ext.profile_enable(False)
    assert "bwd_delta" in prof and "bwd_dq_mfma" in prof and "bwd_mfma" in prof, prof
    assert ext.workspace_stats()["bytes"] >= fast
    ext.set_option("dq", 5)
    try:
        ext.profile_enable(True)
        ref = ext.ex_backward(qd, kd, vd, o, dod, lse, causal, scale)
        torch.cuda.synchronize()
        prof5 = ext.profile_report()
        ext.profile_enable(False)
    finally:
        ext.set_option("dq", 0)
    assert "bwd_delta" not in prof5, prof5
    for a, b in zip(grads, ref):
        assert torch.isfinite(a.float()).all() and max_abs(a, b) <= 2e-2 * max(1.0, float(b.float().abs().max()))
    rq, rk, rv, ro, rlse = orc.extended_attention_backward(q[:2], k[:2], v[:2], do[:2], causal=causal, softmax_scale=scale)
    tol = dtype_tolerances(torch.bfloat16)
    for a, b in zip(grads, (rq, rk, rv)):
        torch.testing.assert_close(a[:2].cpu(), b, **tol)
    # and the last unit (the chunk loop's offsets into the key-side tensors)
    rq2, rk2, rv2, _, _ = orc.extended_attention_backward(q[-1:], k[-1:], v[-1:], do[-1:], causal=causal, softmax_scale=scale)
    for a, b in zip(grads, (rq2, rk2, rv2)):
        torch.testing.assert_close(a[-1:].cpu(), b, **tol)
    ext.release_workspace()


def test_extended_handover_in_chunks(device):
    """The same with the chunk bound lowered so that the (b,h) units are worked through in three chunks: the key-side tensors
    (k, v, dk, dv: Nk rows) and the query-side ones (q, dO, dq: Nq rows) advance by different strides."""
    import flashattention_lab_cuda as ext

    bh, nq, nk, d = 72, 1024, 1536, 128
    g = torch.Generator().manual_seed(99)
    q = torch.randn((bh, nq, d), generator=g).to(torch.bfloat16).to(device)
    k = torch.randn((bh, nk, d), generator=g).to(torch.bfloat16).to(device)
    v = torch.randn((bh, nk, d), generator=g).to(torch.bfloat16).to(device)
    do = torch.randn((bh, nq, d), generator=g).to(torch.bfloat16).to(device)
    scale = d ** -0.5
    o, lse = ext.ex_forward(q, k, v, False, scale)
    one = ext.ex_backward(q, k, v, o, do, lse, False, scale)
    per_unit_mb = (nq // 32) * (8 * (nk // 256)) * 2048 / 2 ** 20      # 3 MiB
    ext.set_option("ds_chunk_mb", int(per_unit_mb * 24))               # 24 units per chunk
    try:
        ext.profile_enable(True)
        three = ext.ex_backward(q, k, v, o, do, lse, False, scale)
        torch.cuda.synchronize()
        prof = ext.profile_report()
        ext.profile_enable(False)
    finally:
        ext.set_option("ds_chunk_mb", 0)
    assert prof["bwd_mfma"][0] == 3 and prof["bwd_dq_mfma"][0] == 3 and prof["bwd_delta"][0] == 1, prof
    for a, b in zip(one, three):
        assert torch.equal(a, b)                                       # the same kernels on the same units
    ext.release_workspace()
