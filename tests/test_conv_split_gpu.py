"""The split-operand convolution path (ymk_conv_split.hip: fp32 operands cut into bf16 planes or two scaled fp16 planes, fp32 accumulation)
against a float64 reference and against the exact fp32-MFMA kernel, on shapes that exercise every gather the fp32 kernel
has: 3x3 with padding / stride / dilation, 1x1, channel counts that are no multiple of 32, residual + activation
epilogues, M tails.  Three bf16 planes (6 MFMAs) and two fp16 planes (3 MFMAs) must be fp32-grade; two bf16 planes (3 MFMAs) within 2^-14 of the
largest output."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # n, h, w, cin, cout, k, stride, pad, dil, act, residual
    (3, 96, 160, 64, 128, 3, 1, 1, 1, "relu", False),   # (every case fills the chip: >= 256 tiles of 128 x 128, or the split path is not taken)
    (8, 195, 323, 96, 192, 3, 2, 1, 1, "none", False),
    (1, 100, 74 * 4, 256, 256, 3, 1, 2, 2, "relu", True),
    (1, 1, 40000, 192, 576, 1, 1, 0, 1, "gelu", False),
    (1, 1, 33000, 72, 64, 1, 1, 0, 1, "none", True),
    (8, 80, 80, 256, 64, 3, 1, 1, 1, "silu", False),
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
            _lib.debug_option("conv_split", split)  # (0 = exact fp32 everywhere; -1 afterwards = unset)
            outs[split] = hipops.conv2d(x.to(dev), wt, sc, bi, r.to(dev) if res else None, stride, pad, dil, act).cpu().double()
    finally:
        _lib.debug_option("conv_split", -1)
    err = {s: float((o - ref).abs().max()) / scale for s, o in outs.items()}
    print(case, {s: f"{e:.2e}" for s, e in err.items()})
    assert err[0] < 3e-6  # a k-ordered fmaf chain of up to 2304 terms
    assert err[3] < 4e-6, "three bf16 planes (6 MFMAs) must be fp32-grade"
    assert err[16] < 4e-6, "two fp16 planes (3 MFMAs): dropped terms are 2^-21 per product, below the fp32 accumulation's rounding"
    assert err[2] < 2.0 ** -14, "two bf16 planes (3 MFMAs): dropped terms are 2^-16 per product"
    again = None
    try:
        _lib.debug_option("conv_split", 3)
        again = hipops.conv2d(x.to(dev), wt, sc, bi, r.to(dev) if res else None, stride, pad, dil, act).cpu().double()
    finally:
        _lib.debug_option("conv_split", -1)
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
        _lib.debug_option("conv_split", -1)
    assert torch.equal(y1, y0 * 2.0 ** (shift - 7))
    far = torch.ones_like(ref, dtype=torch.bool)
    far[1, :, 3:8, 5:10] = False  # outputs that do not see the outlier
    err = float(((y0.double() - ref).abs() * far).max()) / float((ref.abs() * far).max())
    assert err < 4e-6, err


def test_producer_records_bound_the_true_maximum_on_whole_pages(dev):
    """The fp16-split kernels take their scale from the max|x| record the PRODUCER of their input left behind (conv
    epilogues; static bounds for LayerNorm outputs; a source's record for pooled / up-sampled / attention-weighted
    copies), and only fall back to a pass over the input without one.  A stale or incomplete record would overflow the fp16
    planes.  In `amax_check` mode every launch that received a record also measures its input: no record may lie below the
    truth, and none may be so loose (> 2^8) that low-plane bits are lost - over whole pages through all four nets."""
    from yomitoku_amd import DocumentAnalyzer, _lib
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict, synthetic_page_with_truth
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    lite = {"ocr": {"text_detector": {"from_pretrained": False},
                    "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True, "batch_bucketing": True, "source_downscale": True}},
            "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}}}
    an = DocumentAnalyzer(configs=lite, device="cuda:0")
    an.text_detector.model.load_state_dict(dbnet_state_dict(1234, out_bias=-2.0))
    an.text_recognizer.model.load_state_dict(parseq_state_dict(1235, eos_bias=6.0))
    an.layout.layout_parser.model.load_state_dict(rtdetr_state_dict(1240, num_classes=6, score_bias=-2.0))
    an.layout.table_structure_recognizer.model.load_state_dict(rtdetr_state_dict(1243, num_classes=3, score_bias=-1.0))
    pages = [synthetic_page_with_truth(100 + i, *((1600, 1200) if i % 2 else (1200, 1600)))[0] for i in range(8)]
    before = _lib.amax_check_counters()
    try:
        _lib.debug_option("amax_check", int(__import__("os").environ.get("YMK_AMAX_CHECK_LEVEL", 1)))
        res = an.serve(pages, wave=8, in_flight=1)
    finally:
        _lib.debug_option("amax_check", 0)
        an.close()
    assert not any(isinstance(r, BaseException) for r in res)
    checked, below, loose, worst = (a - b for a, b in zip(_lib.amax_check_counters(), before))
    print("records checked", checked, "below the truth", below, "looser than 2^8", loose, "largest exponent distance", _lib.amax_check_counters()[3])
    assert checked >= 100, checked  # most launches of the detector and the RT-DETRv2 backbones, the ViT blocks' GEMMs
    assert below == 0 and loose == 0


DMA_CASES = CASES + [  # n, h, w, cin, cout, k, stride, pad, dil, act, residual
    (1, 1, 70001, 32, 128, 1, 1, 0, 1, "none", False),     # one K tile, M tail
    (1, 1, 66000, 64, 256, 1, 1, 0, 1, "relu", True),      # two K tiles (shorter than the three-stage pipeline)
    (1, 1, 40000, 192, 7119, 1, 1, 0, 1, "none", False),   # ragged Cout (the vocabulary head)
    (2, 100, 148, 512, 512, 3, 1, 2, 2, "relu", False),    # long K, dilation 2
    (6, 203, 331, 128, 64, 3, 2, 1, 1, "relu", False),     # 64-column tile, stride 2, odd sizes
    (2, 160, 160, 4 * 9, 96, 3, 1, 1, 1, "none", True),    # 36 channels: a partial channel tile in every tap
]


@pytest.mark.parametrize("tile", [20, 21])  # 256-row tiles / 8 waves / three stages; 128-row tiles / 4 waves / two stages
@pytest.mark.parametrize("case", DMA_CASES)
def test_f16_split_lds_dma_kernel_equals_the_register_staged_kernel(dev, case, tile):
    """conv_f16_dma (ymk_conv_dma.hip: both operands by LDS-DMA, fp32 activations converted at the fragment read, three LDS
    stages, 256-row tiles) multiplies the same planes and adds each accumulator's terms in the same order as
    conv_igemm_split<FMT = 1>: every output bit must agree - which proves the swizzled source addressing, the zero fill of
    padding taps / channel tails / rows past M by out-of-range DMA lanes, and the pipeline's waits - and both stay fp32-grade
    against float64."""
    from yomitoku_amd import _lib
    from tests import hipops

    n, h, w, cin, cout, k, stride, pad, dil, act, res = case
    g = torch.Generator().manual_seed(17)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    sc, bi = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, pad, dil) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    ref = {"relu": torch.relu, "none": lambda t: t, "gelu": torch.nn.functional.gelu, "silu": torch.nn.functional.silu}[act](ref)
    try:
        _lib.debug_option("conv_split", 16)
        _lib.debug_option("conv_split_tile", 3)  # the register-staged kernel, whatever the automatic choice for this shape
        want = hipops.conv2d(x.to(dev), wt, sc, bi, r.to(dev) if res else None, stride, pad, dil, act).cpu()
        _lib.debug_option("conv_split_tile", tile)
        got = hipops.conv2d(x.to(dev), wt, sc, bi, r.to(dev) if res else None, stride, pad, dil, act).cpu()
        again = hipops.conv2d(x.to(dev), wt, sc, bi, r.to(dev) if res else None, stride, pad, dil, act).cpu()
    finally:
        _lib.debug_option("conv_split", -1)
        _lib.debug_option("conv_split_tile", 0)
    err = float((got.double() - ref).abs().max()) / float(ref.abs().max())
    print(case, f"{err:.2e}", "bit-equal" if torch.equal(got, want) else f"max diff {float((got - want).abs().max()):.3e}")
    assert err < 4e-6
    assert torch.equal(got, again), "bit-identical on repeat"
    assert torch.equal(got, want), float((got - want).abs().max())
