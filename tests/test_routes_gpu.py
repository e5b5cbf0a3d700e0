"""The round-5 routes of the fp16-split path, each against its round-4 form on the same inputs (same planes, same products,
same accumulation order wherever the arithmetic is meant to be identical; fp32-grade agreement where a scale differs):

  * "astat": chip-filling pointwise layers with K <= 256 on the A-stationary kernel (ymk_conv_astat.hip) - the same bits as
    the register-staged kernel (tests/test_conv_astat_gpu.py checks the operator; here: whole nets, routing on / off);
  * "parseq_no_ln_fusion" / "parseq_no_mlp_fusion": the ViT blocks' LayerNorms folded into the operand load of q|k|v and fc1,
    and the whole MLP half (norm2 -> fc1 -> GELU -> fc2 -> + residual) as one launch (models/layers/parseq_transformer.py:
    188-204) against LayerNorm and the two GEMMs as launches of their own;
  * "act_planes": the tensor between a bottleneck's 1 x 1 reduction and its 3 x 3 convolution stored as fp16 planes under a
    bound (models/dbnet_plus.py:33-38, rtdetr_backbone.py) against fp32 activations.

ymk_stat counters prove that the route under test was really taken."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Option:
    def __init__(self, key, value, restore):
        self.key, self.value, self.restore = key, value, restore

    def __enter__(self):
        from yomitoku_amd import _lib

        _lib.debug_option(self.key, self.value)

    def __exit__(self, *exc):
        from yomitoku_amd import _lib

        _lib.debug_option(self.key, self.restore)
        return False


def test_dbnet_with_and_without_fp16_planes_between_reduction_and_3x3(dev):
    from oracle.dbnet import dbnet_forward
    from yomitoku_amd import _lib
    from yomitoku_amd.nets import DBNet
    from yomitoku_amd.utils.synth import dbnet_state_dict

    sd = dbnet_state_dict(1234)
    net = DBNet().load_state_dict(sd).to(dev)
    x = torch.randn(2, 3, 960, 1280, generator=torch.Generator().manual_seed(41))  # layer1 and layer2 fill the chip: 7 pairs
    w0, r0 = _lib.stat("planes_written_launches"), _lib.stat("planes_read_launches")
    with_planes = net(x.to(dev))["binary"].cpu()
    written, read = _lib.stat("planes_written_launches") - w0, _lib.stat("planes_read_launches") - r0
    assert written == read and written >= 7, (written, read)  # layer1 (3) + layer2 (4) at this size; all 16 at the bench's
    assert torch.equal(net(x.to(dev))["binary"].cpu(), with_planes)  # bit-identical on repeat
    with _Option("act_planes", 0, 1):
        w1 = _lib.stat("planes_written_launches")
        without = net(x.to(dev))["binary"].cpu()
        assert _lib.stat("planes_written_launches") == w1
    ref = dbnet_forward(sd, x)["binary"]
    e_with, e_without = (with_planes - ref).abs().max().item(), (without - ref).abs().max().item()
    print("planes", written, "max|dP| vs oracle with / without planes", e_with, e_without, "between them", (with_planes - without).abs().max().item())
    assert e_with < 1e-3 and e_without < 1e-3
    assert e_with < max(4.0 * e_without, 2e-5)  # the bound-derived scale costs no accuracy worth the name
    assert (with_planes - without).abs().max().item() < 5e-5


def test_rtdetr_with_and_without_fp16_planes(dev):
    from tests.test_rtdetr_gpu import _net, assert_same_detections
    from yomitoku_amd import _lib
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    sd = rtdetr_state_dict(1242, num_classes=6)
    net = _net(dev, sd, 6)
    x = torch.rand(4, 3, 640, 640, generator=torch.Generator().manual_seed(9))
    w0 = _lib.stat("planes_written_launches")
    a = net(x.to(dev))
    assert _lib.stat("planes_written_launches") - w0 >= 3
    with _Option("act_planes", 0, 1):
        b = net(x.to(dev))
    assert_same_detections(a["pred_logits"].cpu().numpy(), a["pred_boxes"].cpu().numpy(), b["pred_logits"].cpu().numpy(),
                           b["pred_boxes"].cpu().numpy())


def test_parseq_layernorm_fusion_and_astat_routing_on_and_off(dev):
    from tests.test_parseq_gpu import _net
    from yomitoku_amd import _lib
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    sd = parseq_state_dict(1235, eos_bias=5.5)
    _, net = _net(dev, sd)
    x = synthetic_line_batch(29, 128, 256).to(dev)  # 32 768 token rows: every encoder GEMM fills the chip
    keys = ("astat_launches", "ln_fused_launches", "mlp_fused_launches")
    before = {k: _lib.stat(k) for k in keys}
    fused = net(x).cpu()
    steps = net.last_ar_steps
    n = {k: _lib.stat(k) - before[k] for k in keys}
    # twelve blocks: norm1 inside qkv's operand load, the MLP half as one launch, proj (and K|V, the head) on the A-stationary kernel
    assert n["ln_fused_launches"] == 12 and n["mlp_fused_launches"] == 12 and n["astat_launches"] >= 24, n
    with _Option("parseq_no_mlp_fusion", 1, 0):
        f1 = _lib.stat("ln_fused_launches")
        three_launches = net(x).cpu()
        assert _lib.stat("ln_fused_launches") - f1 == 24 and net.last_ar_steps == steps  # norm2 now inside fc1's operand load
    with _Option("parseq_no_mlp_fusion", 1, 0), _Option("parseq_no_ln_fusion", 1, 0):
        f2, m2 = _lib.stat("ln_fused_launches"), _lib.stat("mlp_fused_launches")
        unfused = net(x).cpu()
        assert _lib.stat("ln_fused_launches") == f2 and _lib.stat("mlp_fused_launches") == m2 and net.last_ar_steps == steps
        with _Option("astat", 0, 1):
            a1 = _lib.stat("astat_launches")
            staged = net(x).cpu()
            assert _lib.stat("astat_launches") == a1 and net.last_ar_steps == steps
    # routing alone changes no bit (same planes, same products, same order per accumulator); the fusions change the
    # normalised rows by the rounding of one fp32 expression and the hidden planes' scale (a bound instead of the maximum)
    assert torch.equal(unfused, staged), (unfused - staged).abs().max().item()
    for other in (three_launches, unfused):
        assert torch.equal(fused.argmax(-1), other.argmax(-1))
        assert (fused - other).abs().max().item() < 1e-4


def test_row_max_head_on_every_kernel_it_can_run_on_changes_no_bit(dev):
    """"rowmax_tile": the greedy loop's vocabulary head (1300 rows x 7119 columns x K = 192, (max, column) per 64-column sub-tile
    in its epilogue) on the A-stationary kernel (0: the default since round 6; 4: its column blocks dealt to groups) and on
    128 x 128 / 256 x 128 tiles (1-3) against the 128 x 64 tile of rounds 3-5 (6) - the same planes, products and K order per
    accumulator and the same pair table, so tokens, step counts and the refined logits are the same bits."""
    from tests.test_parseq_gpu import _net
    from yomitoku_amd import _lib
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    sd = parseq_state_dict(1235, eos_bias=5.5)
    _, net = _net(dev, sd)
    x = synthetic_line_batch(31, 1300, 64).to(dev)
    outs = {}
    for tile in (6, 0, 1, 2, 3, 4):
        with _Option("rowmax_tile", tile, 0):
            w0, a0 = _lib.stat("rowmax_wide_launches"), _lib.stat("astat_launches")
            outs[tile] = (net(x).cpu(), net.last_ar_steps)
            wide, astat = _lib.stat("rowmax_wide_launches") - w0, _lib.stat("astat_launches") - a0
            assert (wide == 0) if tile == 6 else (wide >= outs[tile][1]), (tile, wide)
            if tile in (0, 4):
                assert astat - astat_other >= outs[tile][1], (tile, astat, astat_other)  # the head's launches on top of the encoder's
            else:
                astat_other = astat
    assert outs[6][1] >= 2
    for tile in (0, 1, 2, 3, 4):
        assert outs[tile][1] == outs[6][1] and torch.equal(outs[tile][0], outs[6][0]), tile


def test_whole_pages_with_every_route_on_and_off(dev):
    """Four pages through DocumentAnalyzer.serve with the three routes on (the default) and off: every discrete leaf equal,
    scores to 1e-4 - the same bar `serve == __call__` is held to."""
    from tests.test_pipeline_gpu import _assert_same_schema
    from tests.test_serving_gpu import _analyzer
    from yomitoku_amd import _lib
    from yomitoku_amd.utils.synth import synthetic_page_with_truth

    an = _analyzer()
    pages = [synthetic_page_with_truth(70 + i, 1200, 1600)[0] for i in range(4)]
    c0 = {k: _lib.stat(k) for k in ("astat_launches", "ln_fused_launches", "planes_read_launches")}  # (four pages: below the fused MLP's 256 row blocks)
    on = [r.model_dump() for r in an.serve(pages, wave=4, in_flight=1)]
    assert all(_lib.stat(k) > v for k, v in c0.items()), c0
    with _Option("astat", 0, 1), _Option("parseq_no_ln_fusion", 1, 0), _Option("parseq_no_mlp_fusion", 1, 0), _Option("act_planes", 0, 1):
        off = [r.model_dump() for r in an.serve(pages, wave=4, in_flight=1)]
    an.close()
    assert sum(len(p["words"]) for p in on) > 0
    for a, b in zip(on, off):
        _assert_same_schema(a, b)
