"""Per-kernel parity of the transformer pieces against plain PyTorch fp32 (CPU)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    a, b = a.float().cpu(), b.float().cpu()
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    assert err <= tol, f"max abs err {err}"


@pytest.mark.parametrize("rows,d,eps", [(7, 192, 1e-6), (130, 512, 1e-5), (33, 768, 1e-6), (5, 256, 1e-5)])
def test_layernorm(dev, rows, d, eps):
    from tests import hipops

    g = torch.Generator().manual_seed(d)
    x = torch.randn(rows, d, generator=g) * 3 + 1
    w, b = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    _close(hipops.layernorm(x.to(dev), w, b, eps), F.layer_norm(x, (d,), w, b, eps), 2e-5)


def _ref_attn(q, k, v, heads, mask=None, kpm=None):
    b, lq, d = q.shape
    hd = d // heads
    qh = q.view(b, lq, heads, hd).transpose(1, 2)
    kh = k.view(b, -1, heads, hd).transpose(1, 2)
    vh = v.view(b, -1, heads, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * hd**-0.5
    if mask is not None:
        s = s.masked_fill(mask[None, None], float("-inf"))
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(b, lq, d)


@pytest.mark.parametrize("b,heads,hd,lq,lk", [(2, 6, 32, 184, 184), (1, 8, 64, 100, 100), (3, 8, 96, 40, 40),
                                              (2, 6, 32, 800, 800), (2, 8, 32, 101, 37), (1, 8, 32, 300, 300),
                                              (2, 8, 48, 100, 100), (1, 8, 48, 77, 131)])  # 48: parseq-small (padded to 64 in LDS)
def test_flash_attention(dev, b, heads, hd, lq, lk):
    from tests import hipops

    g = torch.Generator().manual_seed(lq * 7 + hd)
    d = heads * hd
    q, k, v = (torch.randn(b, n, d, generator=g) for n in (lq, lk, lk))
    q = q * 2.0  # sharper softmax: exercises the running-max rescale
    _close(hipops.attention(q.to(dev), k.to(dev), v.to(dev), heads), _ref_attn(q, k, v, heads), 2e-5)


@pytest.mark.parametrize("b,heads,hd,lq,lk", [(2, 6, 32, 184, 184), (1, 8, 64, 100, 100), (3, 8, 96, 40, 40), (2, 6, 32, 800, 800),
                                              (2, 8, 32, 101, 37), (2, 8, 48, 100, 100), (1, 8, 96, 400, 400)])
@pytest.mark.parametrize("gain", [1.0, 300.0, 1e-3])
def test_flash_attention_fp16_split_products(dev, b, heads, hd, lq, lk, gain):
    """k_flash_attn_f16: both products of the attention as fp16-split MFMAs (two scaled fp16 planes per fp32 operand, three
    MFMAs per 16-k step, fp32 accumulate; P in (0, 1] scaled by 2^14) - what the encoders run when their convolutions do.  The
    same tolerance as the exact-fp32 kernel against the float64-free torch reference, at operand magnitudes 1e-3 ... 300
    (the scales follow max|x| of q, k and v: `gain` scales k and v, q is scaled back so that the logits stay the same)."""
    from yomitoku_amd import _lib
    from tests import hipops

    g = torch.Generator().manual_seed(lq * 7 + hd)
    d = heads * hd
    q, k, v = (torch.randn(b, n, d, generator=g) for n in (lq, lk, lk))
    q, k, v = q * 2.0 / gain, k * gain, v * gain
    ref = _ref_attn(q.double(), k.double(), v.double(), heads).float()
    try:
        _lib.debug_option("conv_split", 16)
        got = hipops.attention(q.to(dev), k.to(dev), v.to(dev), heads)
        again = hipops.attention(q.to(dev), k.to(dev), v.to(dev), heads)
    finally:
        _lib.debug_option("conv_split", -1)
    exact = hipops.attention(q.to(dev), k.to(dev), v.to(dev), heads)
    assert torch.equal(got, again)
    scale = float(ref.abs().max())
    e_split, e_exact = float((got.cpu() - ref).abs().max()) / scale, float((exact.cpu() - ref).abs().max()) / scale
    print(f"hd {hd} lq {lq} lk {lk} gain {gain}: fp16-split {e_split:.2e}, exact fp32 {e_exact:.2e}")
    assert e_split < 2e-5 and e_split < 4 * e_exact + 2e-6


@pytest.mark.parametrize("b,heads,hd,lq,lk", [(3, 6, 32, 1, 9), (2, 8, 64, 101, 18), (4, 6, 32, 1, 736), (2, 8, 96, 101, 101)])
def test_small_attention_with_masks(dev, b, heads, hd, lq, lk):
    from tests import hipops

    g = torch.Generator().manual_seed(lk)
    d = heads * hd
    q, k, v = (torch.randn(b, n, d, generator=g) for n in (lq, lk, lk))
    mask = torch.triu(torch.ones(lq, lk, dtype=torch.bool), 1) if lq > 1 else None
    if mask is not None:
        mask[:2] = False
    kpm = torch.zeros(b, lk, dtype=torch.bool)
    for i in range(b):
        kpm[i, max(1, lk - 1 - 3 * i):] = True
    out = hipops.attention(q.to(dev), k.to(dev), v.to(dev), heads, mask_qk=mask, key_padding_mask=kpm)
    _close(out, _ref_attn(q, k, v, heads, mask, kpm), 2e-5)
