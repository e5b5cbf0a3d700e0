"""The A-stationary fp16-split 1 x 1 kernel (yomitoku_amd/csrc/ymk_conv_astat.hip: the models' short-K pointwise layers since
round 5; ymk_op_conv1x1_astat runs it at any row count) against (a) an fp64 product on the CPU, with the tolerance of the other
fp16-split kernels, (b) the register-staged fp16-split kernel (ymk_op_conv2d under "conv_split" 16, "conv_split_tile" 3), bit
for bit - the arithmetic is the same, only the schedule differs - and (c) the library's own routing: the same layer through
ymk_op_conv2d with the automatic tile choice must land on this kernel and give the same bits again.  With a LayerNorm folded
into the operand load: against LayerNorm-then-GEMM in fp64 and against ymk_op_layernorm followed by the unfused kernel."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

M = 33000  # 258 row blocks of 128: the library's own fp16 path takes every case (>= 256 tiles), and the last block is ragged

# (C, Cout, act, scale, bias, residual)
CASES = [
    (192, 576, "none", False, True, False),   # PARSeq-tiny qkv: six K tiles, 4.5 column blocks
    (64, 256, "relu", True, True, True),      # ResNet layer1 expand
    (128, 512, "relu", True, True, True),     # layer2 expand
    (192, 768, "gelu", False, True, False),   # ViT fc1
    (256, 192, "silu", True, False, False),   # eight K tiles: the 64-column form
    (96, 100, "none", False, True, True),     # ragged Cout, three K tiles
    (32, 64, "relu", True, True, False),      # one K tile, one narrow column block
    (72, 130, "sigmoid", False, False, False),  # C not a multiple of 32: the K padding is read as zeros
]


def _reference(x, w, scale, bias, res, act):
    y = x.double().cpu() @ w.double().t()
    if scale is not None:
        y = y * scale.double()
    if bias is not None:
        y = y + bias.double()
    if res is not None:
        y = y + res.double().cpu()
    if act == "relu":
        y = y.clamp(min=0)
    elif act == "gelu":
        y = torch.nn.functional.gelu(y)
    elif act == "silu":
        y = torch.nn.functional.silu(y)
    elif act == "sigmoid":
        y = torch.sigmoid(y)
    return y


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}to{c[1]}_{c[2]}")
def test_astat_matches_fp64_and_the_library_kernels(dev, case):
    from tests import hipops
    from yomitoku_amd import _lib

    c, cout, act, use_scale, use_bias, use_res = case
    g = torch.Generator().manual_seed(c * 1000 + cout)
    x = (torch.randn(M, c, generator=g) * 3.0).to(dev)
    w = torch.randn(cout, c, generator=g) / c ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5 if use_scale else None
    bias = torch.randn(cout, generator=g) if use_bias else None
    res = torch.randn(M, cout, generator=g).to(dev) if use_res else None
    y, ms = hipops.conv1x1_astat(x, w, scale, bias, res, act)
    assert ms > 0
    ref = _reference(x, w, scale, bias, res, act)
    err = (y.double().cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < 3e-6, err
    # the kernels the models run, on the same operands: NCHW views of the same rows
    lib = _lib.load()
    outs = {}
    try:
        _lib.debug_option("conv_split", 16)
        for tile in (3, 0):  # the register-staged kernel, then the library's own choice for this shape
            _lib.debug_option("conv_split_tile", tile)
            _lib.check(lib.ymk_prof_begin())
            y_lib = hipops.conv2d(x.t().reshape(1, c, 1, M), w.reshape(cout, c, 1, 1), scale, bias,
                                  res.t().reshape(1, cout, 1, M) if res is not None else None, act=act)
            ms_l, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            _lib.check(lib.ymk_prof_end(ctypes.byref(ms_l), ctypes.byref(fl), ctypes.byref(ln)))
            rows = _lib.prof_launch_table()
            assert len(rows) == 1 and rows[0][3] == 3.0, rows  # the library did take its fp16-plane path
            outs[tile] = y_lib.reshape(cout, M).t()
    finally:
        _lib.debug_option("conv_split", -1)
        _lib.debug_option("conv_split_tile", 0)
    if cout % 4 == 0:  # the register-staged kernel's 16-byte epilogue: the same expression per value
        assert torch.equal(y, outs[3]), (y - outs[3]).abs().max().item()
    else:  # ragged Cout: that kernel stores through another epilogue form; same values to rounding
        assert (y - outs[3]).abs().max().item() <= 2e-6 * max(1.0, outs[3].abs().max().item())
    if cout > 64:  # the routing rule of conv2d_split: wide short-K pointwise layers run on THIS kernel
        assert torch.equal(y, outs[0]), (y - outs[0]).abs().max().item()
    else:  # 64-column layers stay on the LDS-DMA form: the same planes in the same order
        assert torch.equal(outs[0], outs[3])


# (M, C, Cout, act, bias, residual): ragged and tiny launches - a single row, a second row block with two rows, four
# channels, more column groups than column blocks can fill, the vocabulary head's width
EDGE_CASES = [
    (1, 192, 576, "none", True, False),
    (130, 64, 256, "relu", True, True),
    (127, 4, 8, "none", False, False),
    (200, 192, 7119, "none", True, False),
    (4099, 256, 64, "relu", True, True),
]


@pytest.mark.parametrize("case", EDGE_CASES, ids=lambda c: f"{c[0]}x{c[1]}to{c[2]}")
def test_astat_edge_shapes_match_fp64(dev, case):
    from tests import hipops

    m, c, cout, act, use_bias, use_res = case
    g = torch.Generator().manual_seed(m + c + cout)
    x = torch.randn(m, c, generator=g).to(dev)
    w = torch.randn(cout, c, generator=g) / c ** 0.5
    bias = torch.randn(cout, generator=g) if use_bias else None
    res = torch.randn(m, cout, generator=g).to(dev) if use_res else None
    y, _ = hipops.conv1x1_astat(x, w, None, bias, res, act)
    assert y.shape == (m, cout) and torch.isfinite(y).all()
    ref = _reference(x, w, None, bias, res, act)
    err = (y.double().cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < 3e-6, err


def test_astat_refuses_what_it_cannot_run(dev):
    from tests import hipops
    from yomitoku_amd._lib import YmkError

    x = torch.randn(64, 288).to(dev)  # K = 288 > 256
    with pytest.raises(YmkError):
        hipops.conv1x1_astat(x, torch.randn(32, 288))
    with pytest.raises(YmkError):
        hipops.conv1x1_astat(torch.randn(64, 30).to(dev), torch.randn(32, 30))  # channels not a multiple of 4


LN_CASES = [  # (C, Cout, act, bias): the two fused layers of a ViT block at the widths the kernel carries a LayerNorm for
    (192, 576, "none", True),
    (192, 768, "gelu", True),
    (128, 384, "none", False),
]


@pytest.mark.parametrize("case", LN_CASES, ids=lambda c: f"ln{c[0]}to{c[1]}_{c[2]}")
@pytest.mark.parametrize("rows", [M, 300])
def test_astat_with_the_layernorm_folded_in(dev, case, rows):
    """y = act(LayerNorm(x) . W^T + b) in one launch: against fp64, and against ymk_op_layernorm followed by the same kernel
    without the fusion (the normalised rows differ by the rounding of one fp32 expression, the products by nothing else).
    Rows with a large common offset and one loud channel: what a ViT's residual stream looks like."""
    from tests import hipops

    c, cout, act, use_bias = case
    g = torch.Generator().manual_seed(c + cout + rows)
    x = torch.randn(rows, c, generator=g) * 2.0 + 5.0
    x[:, 7] *= 60.0
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    w = torch.randn(cout, c, generator=g) / c ** 0.5
    bias = torch.randn(cout, generator=g) if use_bias else None
    y, ms = hipops.conv1x1_astat(x.to(dev), w, None, bias, None, act, ln=(gamma, beta, 1e-6))
    assert ms > 0
    xn = torch.nn.functional.layer_norm(x.double(), (c,), gamma.double(), beta.double(), 1e-6)
    ref = _reference(xn, w, None, bias, None, act)
    err = (y.double().cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < 4e-6, err
    xn32 = hipops.layernorm(x.to(dev), gamma, beta, 1e-6)
    y2, _ = hipops.conv1x1_astat(xn32, w, None, bias, None, act)
    assert (y - y2).abs().max().item() <= 4e-6 * max(1.0, ref.abs().max().item())


def test_astat_refuses_a_layernorm_it_cannot_hold(dev):
    from tests import hipops
    from yomitoku_amd._lib import YmkError

    x = torch.randn(256, 256).to(dev)  # K = 256: the planes and a whole fp32 row do not fit the registers
    with pytest.raises(YmkError):
        hipops.conv1x1_astat(x, torch.randn(64, 256), ln=(torch.ones(256), torch.zeros(256), 1e-6))
