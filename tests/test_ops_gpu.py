"""Per-kernel parity: HIP operators (through the C ABI) vs plain PyTorch fp32 on the CPU."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-4  # fp32 MFMA is an exact fmaf chain; the CPU reference sums in another order


def _close(a, b, tol=TOL):
    a = a.float().cpu()
    b = b.float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-6
    assert err <= tol * max(1.0, ref), f"max abs err {err} (ref max {ref})"


def test_gemm_identity_asymmetric(dev):
    """A = I against an asymmetric B catches a transposed C write (guide rule 16)."""
    from tests import hipops

    c = 64
    x = torch.zeros(1, c, 8, 8)
    for i in range(64):
        x[0, i, i // 8, i % 8] = 1.0  # pixel p has one-hot channel p
    w = torch.arange(c * c, dtype=torch.float32).reshape(c, c, 1, 1) / 100.0  # w[co][ci]
    y = hipops.conv2d(x.to(dev), w)
    _close(y, F.conv2d(x, w), 1e-6)


CASES = [
    # (n, cin, h, w, cout, k, stride, pad, dil)
    (1, 64, 24, 40, 64, 1, 1, 0, 1),
    (2, 64, 20, 28, 256, 1, 1, 0, 1),
    (1, 256, 17, 23, 128, 1, 2, 0, 1),
    (1, 64, 19, 21, 64, 3, 1, 1, 1),
    (2, 128, 18, 22, 128, 3, 2, 1, 1),
    (1, 512, 10, 12, 512, 3, 1, 2, 2),
    (1, 32, 30, 34, 32, 3, 1, 1, 1),
    (1, 32, 30, 34, 64, 3, 1, 1, 1),
    (1, 96, 9, 11, 200, 1, 1, 0, 1),      # cin not a multiple of 32, cout ragged
    (1, 4, 5, 7, 36, 1, 1, 0, 1),         # tiny K
    (3, 256, 40, 48, 64, 3, 1, 1, 1),     # wide-M path (128x64 tiles)
    (1, 1024, 26, 30, 2048, 1, 1, 0, 1),  # 128x128 tiles
    (1, 512, 20, 20, 512, 3, 1, 1, 1),    # RT-DETR res5: 400 pixels, 144 K tiles (grid-starved, split-K)
    (1, 256, 1, 300, 96, 1, 1, 0, 1),     # decoder linear: 300 queries
    (1, 192, 1, 42, 1000, 1, 1, 0, 1),    # PARSeq head shape: few rows, wide N, 6 K tiles
    (1, 64, 1, 3, 8, 1, 1, 0, 1),         # fewer K tiles than waves
]


# kernel routing: the library's own choice, conv_igemm only, or one fixed split-K shape
# (ymk_conv.hip try_splitk candidates: 64x64/4, 64x32/4, 64x32/8, 32x32/4, 32x32/8 waves)
ROUTES = ([("auto", {}), ("igemm", {"no_splitk": 1})] + [(f"splitk{i}", {"splitk_force": i}) for i in range(5)]
          + [(f"variant{v}", {"no_splitk": 1, "conv_variant": v}) for v in range(1, 7)])
# conv_fast bit 2: accumulators straight to global memory for every plain store; bit 3: swizzled K tiles (three 128 x 64
# blocks per CU); bit 4: direct epilogue where 16-byte stores are impossible (ragged Cout).  27 is the library's default,
# 3 the LDS-staged epilogue / padded K tiles of round 2.
EPILOGUES = [("staged, padded (round 2)", {"conv_fast": 3}), ("direct", {"conv_fast": 7}), ("swizzled", {"conv_fast": 11}),
             ("direct+swizzled", {"conv_fast": 15})]
ROUTES += [(f"{name}, variant{v}", dict(opts, no_splitk=1, conv_variant=v)) for name, opts in EPILOGUES for v in (0, 1, 2, 3, 4, 7, 8)]
RESET = (("splitk_force", -1), ("no_splitk", 0), ("conv_variant", 0), ("conv_fast", 27))


@pytest.mark.parametrize("route", ROUTES, ids=[r[0] for r in ROUTES])
@pytest.mark.parametrize("case", CASES)
def test_conv2d_matches_torch(dev, case, route):
    from yomitoku_amd import _lib
    from tests import hipops

    try:
        for key, val in route[1].items():
            _lib.debug_option(key, val)
        _conv_case(dev, case, hipops)
    finally:
        for key, val in RESET:
            _lib.debug_option(key, val)


@pytest.mark.parametrize("epilogue", EPILOGUES, ids=[e[0] for e in EPILOGUES])
@pytest.mark.parametrize("case", CASES + [(8, 64, 50, 74, 256, 1, 1, 0, 1), (8, 256, 40, 37, 64, 3, 1, 1, 1), (1, 192, 1, 29000, 200, 1, 1, 0, 1),
                                         (1, 192, 1, 12000, 7119, 1, 1, 0, 1)])
def test_conv_epilogue_variants_are_bit_identical(dev, case, epilogue):
    """The direct epilogue and the swizzled K-tile layout change where values travel, not what is computed: every output
    bit under the library's default (swizzled 128 x 64 tiles, direct epilogue for ragged Cout) equals the one under each
    other setting, with and without residual, for every activation (the last cases have enough blocks for the wide tiles)."""
    from yomitoku_amd import _lib
    from tests import hipops

    n, cin, h, w, cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale, bias = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    oh, ow = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = torch.randn(n, cout, oh, ow, generator=g).to(dev)
    configs = [(None, None, None, "none"), (scale, bias, res, "relu"), (scale, None, None, "gelu"), (None, bias, res, "silu"),
               (scale, bias, None, "sigmoid")]
    try:
        for no_splitk in (0, 1):
            _lib.debug_option("no_splitk", no_splitk)
            want = [hipops.conv2d(x, wt, sc, bi, rs, stride, pad, dil, act) for sc, bi, rs, act in configs]
            for key, val in epilogue[1].items():
                _lib.debug_option(key, val)
            got = [hipops.conv2d(x, wt, sc, bi, rs, stride, pad, dil, act) for sc, bi, rs, act in configs]
            _lib.debug_option("conv_fast", _lib.CONV_FAST_DEFAULT)
            for (sc, bi, rs, act), a, b in zip(configs, want, got):
                assert torch.equal(a, b), (act, rs is not None, float((a - b).abs().max()))
    finally:
        for key, val in RESET:
            _lib.debug_option(key, val)


def _conv_case(dev, case, hipops):
    n, cin, h, w, cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % (2**31))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, None, stride, pad, dil) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref + res)
    y = hipops.conv2d(x.to(dev), wt, scale, bias, res.to(dev), stride, pad, dil, "relu")
    _close(y, ref)


def test_stem_conv7x7_tap4(dev):
    from tests import hipops

    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 64, 96, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    y = hipops.conv2d(x.to(dev), wt, None, None, None, 2, 3, 1, "relu")
    _close(y, F.relu(F.conv2d(x, wt, None, 2, 3)))


def test_conv3x3_stem_tap4_s2(dev):
    from tests import hipops

    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 3, 40, 40, generator=g)
    wt = torch.randn(32, 3, 3, 3, generator=g) / 5.0
    y = hipops.conv2d(x.to(dev), wt, None, None, None, 2, 1, 1, "none")
    _close(y, F.conv2d(x, wt, None, 2, 1))


@pytest.mark.parametrize("act", ["silu", "sigmoid", "gelu"])
def test_conv_activations(dev, act):
    from tests import hipops

    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 64, 12, 12, generator=g)
    wt = torch.randn(64, 64, 1, 1, generator=g) / 8.0
    ref = F.conv2d(x, wt)
    ref = {"silu": F.silu, "sigmoid": torch.sigmoid, "gelu": F.gelu}[act](ref)
    _close(hipops.conv2d(x.to(dev), wt, act=act), ref, 1e-5)


def test_maxpool(dev):
    from tests import hipops

    x = torch.randn(2, 64, 33, 47, generator=torch.Generator().manual_seed(1))
    _close(hipops.maxpool3x3s2(x.to(dev)), F.max_pool2d(x, 3, 2, 1), 0.0)


@pytest.mark.parametrize("shape,size", [((1, 64, 10, 14), (20, 28)), ((2, 64, 7, 9), (28, 36)), ((1, 256, 9, 12), (18, 25))])
def test_bilinear(dev, shape, size):
    from tests import hipops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g)
    add = torch.randn(shape[0], shape[1], *size, generator=g)
    ref = F.interpolate(x, size=size, mode="bilinear", align_corners=False) + add
    _close(hipops.upsample_bilinear(x.to(dev), size, add.to(dev)), ref, 1e-6)


def test_prof_launch_table_has_one_row_per_launch(dev):
    """ymk_prof_launch_table (the per-launch rows bench.py prices against the two roofs): as many rows as ymk_prof_end counted,
    FLOPs and bytes summing to the span's totals, and the kernel family of each launch (exact fp32 / fp16 planes)."""
    import ctypes

    from tests import hipops
    from yomitoku_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    x_big = torch.randn(2, 64, 200, 160, generator=g).to(dev)     # 64 000 output pixels x 128 channels: fills the chip
    w_big = torch.randn(128, 64, 3, 3, generator=g) / 24
    x_small = torch.randn(1, 32, 20, 20, generator=g).to(dev)
    w_small = torch.randn(48, 32, 1, 1, generator=g) / 6
    res = torch.randn(1, 48, 20, 20, generator=g).to(dev)
    try:
        _lib.check(lib.ymk_prof_begin())
        hipops.conv2d(x_big, w_big, padding=1)                      # ymk_op_conv2d: exact fp32 unless asked
        _lib.debug_option("conv_split", 16)
        hipops.conv2d(x_big, w_big, padding=1)                      # two fp16 planes
        hipops.conv2d(x_small, w_small, residual=res)               # grid-starved: stays on the exact kernels
        _lib.debug_option("conv_split", -1)
        torch.cuda.synchronize()
        ms, fl, ln, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
        _lib.check(lib.ymk_prof_bytes(ctypes.byref(by)))
    finally:
        _lib.debug_option("conv_split", -1)
    rows = _lib.prof_launch_table()
    assert len(rows) == ln.value == 3
    assert sum(r[1] for r in rows) == pytest.approx(fl.value, rel=1e-12) and sum(r[2] for r in rows) == pytest.approx(by.value, rel=1e-12)
    assert sum(r[0] for r in rows) == pytest.approx(ms.value, rel=1e-6) and all(r[0] > 0 for r in rows)
    flop_big = 2.0 * 2 * 200 * 160 * 128 * 64 * 9
    assert rows[0][1] == rows[1][1] == flop_big and rows[2][1] == 2.0 * 400 * 48 * 32
    bytes_big = 4.0 * (2 * 200 * 160 * 64 + 128 * 64 * 9 + 2 * 200 * 160 * 128)
    assert rows[0][2] == bytes_big and rows[2][2] == 4.0 * (400 * 32 + 48 * 32 + 2 * 400 * 48)   # the residual is read as well
    assert [r[3] for r in rows] == [0.0, 3.0, 0.0]
    # a second span starts from nothing
    _lib.check(lib.ymk_prof_begin())
    _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
    assert _lib.prof_launch_table() == [] and ln.value == 0


def test_gelu_of_the_library_stays_within_its_stated_absolute_error(dev):
    """ymk_common.h gelu_f32 (erf through a rational erfc form, v_rcp + v_exp: every kernel's GELU since round 5, the exact-fp32
    mode included - include/ymk.h says so) against float64 erf over [-12, 12]: the ABSOLUTE error stays below 5e-7 (measured
    4.2e-7; the rounding of v times a libm-grade erff is the same size).  The RELATIVE error in the negative tail is not bounded -
    beyond v = -5 the true value is below 1e-6 and the result is a few 1e-7 of either sign: callers that care about the
    tail's relative accuracy (none on this path: the values feed fp32 sums next to O(1) terms) must not use it."""
    import math

    from tests.hipops import conv2d

    n = 1 << 16
    v = torch.linspace(-12.0, 12.0, n, dtype=torch.float64)
    v = torch.cat([v, torch.tensor([0.0, -0.0, 1e-30, -1e-30, 5.5, -5.5, 8.0, -8.0], dtype=torch.float64)])
    pad = (-v.numel()) % 4
    v = torch.cat([v, torch.zeros(pad, dtype=torch.float64)])
    x = v.float().reshape(1, 4, -1, 1).contiguous()  # N x C x H x W with 4 channels: an identity 1 x 1 convolution + GELU
    w = torch.eye(4).reshape(4, 4, 1, 1)
    got = conv2d(x.to(dev), w, act="gelu").cpu().double().reshape(-1)
    xin = x.double().reshape(-1)
    want = 0.5 * xin * (1.0 + torch.erf(xin / math.sqrt(2.0)))
    err = (got - want).abs()
    assert err.max().item() < 5e-7, err.max().item()
    big = xin.abs() > 9.0
    assert torch.equal(got[big & (xin > 0)], xin[big & (xin > 0)]) and (got[big & (xin < 0)].abs() < 5e-7).all()
    assert got[xin == 0].abs().max().item() == 0.0
