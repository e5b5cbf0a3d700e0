"""The split-operand convolution path (ymk_conv_split.hip: fp32 operands cut into bf16 planes or two scaled fp16 planes, fp32 accumulation)
against a float64 reference and against the exact fp32-MFMA kernel, on shapes that exercise every gather the fp32 kernel
has: 3x3 with padding / stride / dilation, 1x1, channel counts that are no multiple of 32, residual + activation
epilogues, M tails.  Three bf16 planes (6 MFMAs) and two fp16 planes (3 MFMAs) must be fp32-grade; two bf16 planes (3 MFMAs) within 2^-14 of the
largest output."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # n, h, w, cin, cout, k, stride, pad, dil, act, residual
    (2, 96, 160, 64, 128, 3, 1, 1, 1, "relu", False),
    (2, 97, 161, 96, 192, 3, 2, 1, 1, "none", False),
    (1, 100, 74 * 4, 256, 256, 3, 1, 2, 2, "relu", True),
    (1, 1, 40000, 192, 576, 1, 1, 0, 1, "gelu", False),
    (1, 1, 33000, 72, 64, 1, 1, 0, 1, "none", True),
    (4, 80, 80, 256, 64, 3, 1, 1, 1, "silu", False),
]


@pytest.mark.parametrize("case", CASES)
def test_bf16_split_conv_matches_fp64_and_fp32_kernels(dev, case):
    from yomitoku_amd import _lib
    from tests import hipops

    n, h, w, cin, cout, k, stride, pad, dil, act, res = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    sc, bi = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, pad, dil) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    ref = {"relu": torch.relu, "none": lambda t: t, "gelu": torch.nn.functional.gelu, "silu": torch.nn.functional.silu}[act](ref)
    scale = float(ref.abs().max())
    outs = {}
    try:
        for split in (0, 3, 2, 16):
            _lib.debug_option("conv_split", split)
            outs[split] = hipops.conv2d(x.to(dev), wt, sc, bi, r.to(dev) if res else None, stride, pad, dil, act).cpu().double()
    finally:
        _lib.debug_option("conv_split", 0)
    err = {s: float((o - ref).abs().max()) / scale for s, o in outs.items()}
    print(case, {s: f"{e:.2e}" for s, e in err.items()})
    assert err[0] < 2e-6
    assert err[3] < 4e-6, "three bf16 planes (6 MFMAs) must be fp32-grade"
    assert err[16] < 4e-6, "two fp16 planes (3 MFMAs): dropped terms are 2^-21 per product, below the fp32 accumulation's rounding"
    assert err[2] < 2.0 ** -14, "two bf16 planes (3 MFMAs): dropped terms are 2^-16 per product"
    again = None
    try:
        _lib.debug_option("conv_split", 3)
        again = hipops.conv2d(x.to(dev), wt, sc, bi, r.to(dev) if res else None, stride, pad, dil, act).cpu().double()
    finally:
        _lib.debug_option("conv_split", 0)
    assert torch.equal(again, outs[3]), "bit-identical on repeat"


@pytest.mark.parametrize("shift", [-40, -12, 0, 9, 30])
def test_f16_split_follows_the_magnitude_of_its_operands(dev, shift):
    """The fp16 planes hold the operands times a power of two taken from max|x| (activations, per launch) and from the
    row maximum (weights): scaling the input by 2^shift must scale the output by exactly 2^shift - the scaled operands,
    hence the accumulators, are the same bits - and one outlier 2^12 above the rest must not cost the rest its accuracy."""
    from yomitoku_amd import _lib
    from tests import hipops

    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 72, 150, generator=g)
    wt = torch.randn(128, 64, 3, 3, generator=g) / 24.0
    x[1, 3, 5, 7] = 4096.0  # the scale follows the outlier; everything else keeps 11 + 11 bits
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, 1, 1)
    try:
        _lib.debug_option("conv_split", 16)
        y0 = hipops.conv2d(x.to(dev), wt, None, None, None, 1, 1, 1, "none").cpu()
        y1 = hipops.conv2d((x * 2.0 ** shift).to(dev), wt * 2.0 ** -7, None, None, None, 1, 1, 1, "none").cpu()
    finally:
        _lib.debug_option("conv_split", 0)
    assert torch.equal(y1, y0 * 2.0 ** (shift - 7))
    far = torch.ones_like(ref, dtype=torch.bool)
    far[1, :, 3:8, 5:10] = False  # outputs that do not see the outlier
    err = float(((y0.double() - ref).abs() * far).max()) / float((ref.abs() * far).max())
    assert err < 4e-6, err
