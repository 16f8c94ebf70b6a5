"""CPU: the oracle's restatement of the extended attention path (Nq != Nk with the bottom-right causal mask,
dense mask, block-sparse mask, dropout) against the golden vectors the REFERENCE's own notebook code produced
(tests/golden/make_golden_ex.py: MultiHeadAttention._block_sparse_flash_attention + look_ahead_mask_), and the
counter-based dropout generator against an independent big-integer evaluation."""
import pytest
import torch

from oracle import attention_oracle as orc
from tests.helpers import ex_golden_tags, load_ex_golden


def test_fixture_set_is_complete():
    assert len(ex_golden_tags()) == 6


@pytest.mark.parametrize("tag", ex_golden_tags())
def test_extended_oracle_matches_reference_vectors(tag):
    meta, g = load_ex_golden(tag)
    b, h, nq, nk, d = (meta[x] for x in ("b", "h", "nq", "nk", "d"))
    q, k, v = (g[x].reshape(b * h, -1, d) for x in "qkv")
    # the reference folds the causal mask into `mask` (look_ahead_mask_): pass it both ways
    kw = dict(softmax_scale=meta["tau"] / d ** 0.5, block_mask=g["block_mask"], br=meta["br"], bc=meta["bc"])
    o1, _ = orc.extended_attention(q, k, v, mask=g.get("mask"), **kw)
    assert (o1.float() - g["o"].reshape(b * h, nq, d)).abs().max().item() < 2e-5
    if meta["causal"] and not meta["dense_mask"]:
        o2, _ = orc.extended_attention(q, k, v, causal=True, **kw)
        assert (o2.float() - g["o"].reshape(b * h, nq, d)).abs().max().item() < 2e-5


def test_square_causal_without_extras_is_the_plain_oracle():
    g = torch.Generator().manual_seed(1)
    q, k, v, do = (torch.randn((2, 37, 24), generator=g) for _ in range(4))
    a = orc.extended_attention_backward(q, k, v, do, causal=True, softmax_scale=0.3)
    b = orc.exact_attention_backward(q, k, v, do, True, 0.3, math_dtype=torch.float64)   # dq, dk, dv, o, lse
    for x, y in zip(a, b):
        assert (x.double() - y.double()).abs().max().item() < 1e-5


def _splitmix_keep(bh_i, row, key, nq, p, seed):
    """the generator of csrc/fa_ex_common.h in Python integers: one splitmix64 value per 2 x 2 quad, 16 bits per element"""
    m = (1 << 64) - 1
    g = 0x9E3779B97F4A7C15
    ctr = (((bh_i * ((nq + 1) // 2) + row // 2) & 0xFFFFFFFF) << 32) | (key // 2)
    z = (ctr + seed * g + g) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    u = (z >> (16 * (2 * (row & 1) + (key & 1)))) & 0xFFFF
    return u >= int(p * 65536.0) + 1


def test_dropout_generator_known_answers_and_statistics():
    bh, nq, nk, p, seed = 3, 17, 29, 0.25, 123456789
    keep = orc.dropout_keep(bh, nq, nk, p, seed)
    for b_, r_, c_ in ((0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (0, 0, 28), (1, 16, 5), (2, 16, 28), (2, 7, 13)):
        assert bool(keep[b_, r_, c_]) == _splitmix_keep(b_, r_, c_, nq, p, seed)
    # the four fields of one value are used by four different elements, and the value itself is splitmix64's
    assert _splitmix_keep(0, 0, 0, 2, 0.5, 0) == (((0xE220A8397B1DCDAF >> 0) & 0xFFFF) >= 32769)
    big = orc.dropout_keep(8, 256, 256, 0.3, 7).float().mean().item()
    assert abs(big - 0.7) < 5e-3
    assert torch.equal(orc.dropout_keep(2, 8, 8, 0.5, 9), orc.dropout_keep(2, 8, 8, 0.5, 9))
    assert not torch.equal(orc.dropout_keep(2, 8, 8, 0.5, 9), orc.dropout_keep(2, 8, 8, 0.5, 10))
    assert orc.dropout_keep(2, 4, 4, 0.0, 1).all()


def test_rows_without_a_visible_key_and_dropout_scaling():
    g = torch.Generator().manual_seed(2)
    q, k, v = torch.randn((1, 6, 8), generator=g), torch.randn((1, 4, 8), generator=g), torch.randn((1, 4, 8), generator=g)
    o, lse = orc.extended_attention(q, k, v, causal=True)      # Nq > Nk: the first two rows see no key
    assert torch.equal(o[0, :2], torch.zeros_like(o[0, :2])) and torch.isinf(lse[0, :2]).all() and torch.isfinite(lse[0, 2:]).all()
    # E[dropout(P) V] = P V: average over many seeds
    acc = sum(orc.extended_attention(q, k, v, dropout_p=0.4, seed=s)[0] for s in range(400)) / 400
    ref, _ = orc.extended_attention(q, k, v)
    assert (acc - ref).abs().max().item() < 0.25


def test_host_module_argument_handling_without_gpu():
    from common.attention_ex import flash_attention_ex, look_ahead_mask

    m = look_ahead_mask(3, 5)
    assert m.shape == (1, 1, 3, 5) and m[0, 0].tolist() == [[True, True, True, False, False], [True, True, True, True, False],
                                                           [True] * 5]
    assert torch.equal(m[0, 0], orc.extended_visible(1, 3, 5, causal=True)[0])
    with pytest.raises(RuntimeError, match="CUDA tensors"):
        flash_attention_ex(torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 4))


def test_mask_normalisation_follows_ordinary_broadcasting():
    """`mask` is "broadcastable to (B, H, Nq, Nk)" (flashattention_pytorch.py:139-141 indexes it as 4-D and lets
    masked_fill broadcast): key-padding masks with a singleton query dim, per-batch masks with a singleton head dim, shared masks."""
    from common.attention_ex import normalize_mask

    b, h, nq, nk = 2, 3, 5, 7
    g = torch.Generator().manual_seed(0)
    full = torch.rand((b, h, nq, nk), generator=g) > 0.4
    want = lambda m: torch.broadcast_to(m, (b, h, nq, nk)).reshape(b * h, nq, nk)   # noqa: E731
    # shared forms -> (Nq, Nk)
    for m in (full[0, 0], full[:1, :1], full[:1, :1, :1, :], full[0, 0, :1]):
        got = normalize_mask(m, (b, h), nq, nk)
        assert got.shape == (nq, nk) and torch.equal(got, torch.broadcast_to(m, (1, 1, nq, nk))[0, 0])
    # per-(b,h) forms -> (BH, Nq, Nk)
    for m in (full, full[:, :1], full[:1], full[:, :1, :1, :], full[:, :, :, :1], full[0]):   # the last: (H, Nq, Nk), aligned from the right
        got = normalize_mask(m, (b, h), nq, nk)
        assert got.shape == (b * h, nq, nk) and torch.equal(got, want(m)), m.shape
    # merged (BH, N, d) tensors: (BH, Nq, Nk), (1, Nq, Nk), (Nq, Nk), and a 4-D mask whose leading dims multiply to BH
    assert torch.equal(normalize_mask(full.reshape(b * h, nq, nk), (b * h,), nq, nk), full.reshape(b * h, nq, nk))
    assert normalize_mask(full[0, :1], (b * h,), nq, nk).shape == (nq, nk)
    assert torch.equal(normalize_mask(full, (b * h,), nq, nk), full.reshape(b * h, nq, nk))
    assert normalize_mask((full[0, 0]).to(torch.uint8), (b * h,), nq, nk).dtype == torch.bool   # 0 / 1 tensors too
    with pytest.raises(RuntimeError, match="does not broadcast"):
        normalize_mask(full[:, :2], (b, h), nq, nk)
    with pytest.raises(RuntimeError, match="does not broadcast"):
        normalize_mask(full, (5,), nq, nk)
