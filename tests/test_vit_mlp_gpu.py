"""The fused ViT MLP (yomitoku_amd/csrc/ymk_vit_mlp.hip): x + fc2(GELU(fc1(LayerNorm(x)))) in one launch with the hidden state on
chip, against the same expression in float64 (timm's Block, models/layers/parseq_transformer.py:188-204) and, inside the
recogniser, against the three launches it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu

D, F = 192, 768


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.2
    w1, b1 = torch.randn(F, D, generator=g) / D ** 0.5, torch.randn(F, generator=g) * 0.3
    w2, b2 = torch.randn(D, F, generator=g) / F ** 0.5, torch.randn(D, generator=g) * 0.3
    return g, gamma, beta, w1, b1, w2, b2


def _reference(x, gamma, beta, w1, b1, w2, b2, eps=1e-6):
    xd = x.double()
    xn = torch.nn.functional.layer_norm(xd, (D,), gamma.double(), beta.double(), eps)
    return xd + torch.nn.functional.gelu(xn @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()


@pytest.mark.parametrize("rows,kind", [(33000, "plain"), (32641, "offset"), (70001, "outlier")])
def test_fused_mlp_matches_fp64(dev, rows, kind):
    """Ragged last block (33000 = 257 x 128 + 104), the smallest launch the kernel takes, rows with a common offset, one channel
    60 x louder than the rest, and weight rows / columns a few powers of two apart (the per-row plane scales)."""
    from tests import hipops

    g, gamma, beta, w1, b1, w2, b2 = _weights(rows)
    x = torch.randn(rows, D, generator=g) * 1.5
    if kind != "plain":
        x = x + 4.0
    if kind == "outlier":
        x[:, 11] *= 60.0
        w1[5] *= 64.0
        w2[:, 700] *= 32.0
        w2[17] *= 0.03125
    y, ms = hipops.vit_mlp(x.to(dev), gamma, beta, 1e-6, w1, b1, w2, b2)
    assert ms > 0 and y.shape == x.shape and torch.isfinite(y).all()
    ref = _reference(x, gamma, beta, w1, b1, w2, b2)
    err = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(kind, rows, f"max error / max|y| = {err:.2e}", f"{ms * 1e3:.0f} us")
    assert err < 4e-6, err
    y2, _ = hipops.vit_mlp(x.to(dev), gamma, beta, 1e-6, w1, b1, w2, b2)
    assert torch.equal(y, y2)  # bit-identical on repeat


def test_fused_mlp_refuses_other_shapes(dev):
    from tests import hipops
    from yomitoku_amd._lib import YmkError

    g, gamma, beta, w1, b1, w2, b2 = _weights(1)
    with pytest.raises(YmkError):
        hipops.vit_mlp(torch.randn(4000, D).to(dev), gamma, beta, 1e-6, w1, b1, w2, b2)  # fewer row blocks than CUs


def test_recogniser_with_the_mlp_fused_and_as_three_launches(dev):
    """A chip-filling grouped forward of the --lite recogniser: twelve fused launches (ymk_stat), the same tokens and step
    counts as the unfused form, logits within 1e-4 of it and within 1e-3 of the CPU oracle."""
    from oracle.parseq import parseq_forward
    from tests.test_parseq_gpu import _net
    from yomitoku_amd import _lib
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    sd = parseq_state_dict(1235, eos_bias=5.5)
    ocfg, net = _net(dev, sd)
    x = synthetic_line_batch(31, 160, 256)  # 40 960 token rows: 320 row blocks
    n0 = _lib.stat("mlp_fused_launches")
    fused = net(x.to(dev)).cpu()
    steps = net.last_ar_steps
    assert _lib.stat("mlp_fused_launches") - n0 == 12
    _lib.debug_option("parseq_no_mlp_fusion", 1)
    try:
        unfused = net(x.to(dev)).cpu()
        assert _lib.stat("mlp_fused_launches") - n0 == 12 and net.last_ar_steps == steps
    finally:
        _lib.debug_option("parseq_no_mlp_fusion", 0)
    ref, ref_steps = parseq_forward(sd, ocfg, x, return_steps=True)
    print("fused vs unfused", (fused - unfused).abs().max().item(), "fused vs oracle", (fused - ref).abs().max().item(),
          "unfused vs oracle", (unfused - ref).abs().max().item())
    assert steps == ref_steps
    assert torch.equal(fused.argmax(-1), unfused.argmax(-1)) and torch.equal(fused.argmax(-1), ref.argmax(-1))
    assert (fused - unfused).abs().max().item() < 1e-4
    assert (fused - ref).abs().max().item() < 1e-3
