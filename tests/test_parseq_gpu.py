"""PARSeq forward parity: HIP path (ymk_parseq_forward) vs the CPU oracle and the golden vectors the
REFERENCE class produced.  Tolerance (BASELINE.json north_star): text logits within 1e-3."""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _net(dev, sd, preset="parseq-tiny-dynw-v4", **cfg_over):
    from oracle.parseq import PRESETS, make_cfg
    from yomitoku_amd.nets import PARSeq

    ocfg = make_cfg(**{**PRESETS[preset], **cfg_over})
    cfg = {
        "num_tokens": ocfg.num_tokens, "max_label_length": ocfg.max_label_length, "refine_iters": ocfg.refine_iters,
        "decode_ar": 1, "repetition_stop": ocfg.repetition_stop, "data": {"img_size": [32, 800]},
        "encoder": {"patch_size": list(ocfg.patch), "num_heads": ocfg.enc_heads, "embed_dim": ocfg.enc_dim,
                    "mlp_ratio": 4, "depth": ocfg.enc_depth},
        "decoder": {"embed_dim": ocfg.dec_dim, "num_heads": ocfg.dec_heads, "mlp_ratio": 4, "depth": 1},
    }
    return ocfg, PARSeq(cfg).load_state_dict(sd).to(dev)


@pytest.mark.parametrize("tag", ["eos", "rep"])
def test_matches_reference_golden(dev, tag):
    from yomitoku_amd.utils.synth import parseq_state_dict

    z = np.load(os.path.join(GOLD, f"parseq_ref_{tag}.npz"))
    kw = ast.literal_eval(str(z["ckpt"]))  # repr() of the synth.parseq_state_dict kwargs
    sd = parseq_state_dict(**kw)
    _, net = _net(dev, sd)
    logits = net(torch.from_numpy(z["x"]).to(dev))
    assert net.last_ar_steps == int(z["steps"])
    lg = logits.cpu()
    assert lg.shape[:2] == z["ids"].shape
    assert np.array_equal(lg.argmax(-1).numpy().astype(np.int32), z["ids"])
    assert np.abs(lg.max(-1).values.numpy() - z["top"]).max() < LOGIT_TOL
    assert np.abs(lg[:, :, ::97].numpy() - z["sample"]).max() < LOGIT_TOL
    # fused tokenizer statistics == softmax().max()
    ids, probs = net.token_stats(logits)
    ref_p, ref_i = lg.softmax(-1).max(-1)
    assert torch.equal(ids.cpu().long(), ref_i)
    assert (probs.cpu() - ref_p).abs().max().item() < 1e-5


@pytest.mark.parametrize("batch,width,seed", [(5, 160, 1), (2, 800, 2), (9, 72, 3)])
def test_matches_oracle_tiny(dev, batch, width, seed):
    from oracle.parseq import parseq_forward
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    sd = parseq_state_dict(1235, eos_bias=5.5)
    ocfg, net = _net(dev, sd)
    x = synthetic_line_batch(seed, batch, width)
    ref, steps = parseq_forward(sd, ocfg, x, return_steps=True)
    out = net(x.to(dev)).cpu()
    assert net.last_ar_steps == steps
    assert out.shape == ref.shape
    assert torch.equal(out.argmax(-1), ref.argmax(-1))
    assert (out - ref).abs().max().item() < LOGIT_TOL


def test_matches_oracle_open_beta_no_refine(dev):
    """parseq (open-beta) geometry: 8x8 patches, D=512, 8 heads, charset v1; refine_iters=0 returns the
    AR logits of the executed steps only."""
    from oracle.parseq import parseq_forward
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    kw = dict(patch=(8, 8), enc_dim=512, dec_dim=512, num_tokens=7312, enc_depth=3)
    sd = parseq_state_dict(77, eos_bias=6.0, **kw)
    ocfg, net = _net(dev, sd, "parseq", enc_depth=3, refine_iters=0)
    x = synthetic_line_batch(5, 4, 224)
    ref, steps = parseq_forward(sd, ocfg, x, return_steps=True)
    out = net(x.to(dev)).cpu()
    assert out.shape == ref.shape and out.shape[1] == steps
    assert (out - ref).abs().max().item() < LOGIT_TOL


def test_matches_oracle_small_geometry(dev):
    """parseq-small geometry: 16x16 patches, D = 384, 8 heads -> head dim 48 (flash attention pads it to 64 in LDS;
    the decoder takes the per-op path because 48 / 4 lanes per head is not a power of two)."""
    from oracle.parseq import parseq_forward
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    kw = dict(patch=(16, 16), enc_dim=384, dec_dim=384, num_tokens=7312, enc_depth=2)
    sd = parseq_state_dict(78, eos_bias=6.0, **kw)
    ocfg, net = _net(dev, sd, "parseq", patch=(16, 16), enc_dim=384, enc_heads=8, dec_dim=384, dec_heads=8, enc_depth=2)
    x = synthetic_line_batch(6, 3, 800)
    ref, steps = parseq_forward(sd, ocfg, x, return_steps=True)
    out = net(x.to(dev)).cpu()
    assert net.last_ar_steps == steps and out.shape == ref.shape
    assert torch.equal(out.argmax(-1), ref.argmax(-1))
    assert (out - ref).abs().max().item() < LOGIT_TOL


def _groups(seed, shapes):
    from yomitoku_amd.utils.synth import synthetic_line_batch

    return [synthetic_line_batch(seed + i, b, w) for i, (b, w) in enumerate(shapes)]


@pytest.mark.parametrize("shapes", [[(5, 160), (2, 800), (9, 72), (1, 96)], [(3, 64), (3, 64)], [(4, 240)]])
def test_grouped_forward_matches_oracle_and_single_calls(dev, shapes):
    """ymk_parseq_forward_groups: several mini-batches, each with its own padded width, through one forward with one
    greedy loop.  Per group: logits vs the oracle run on that group alone (1e-3, same arg-max), the step count the
    group's own loop would have stopped at, and agreement with the single-group entry point."""
    from oracle.parseq import parseq_forward
    from yomitoku_amd.utils.synth import parseq_state_dict

    sd = parseq_state_dict(1235, eos_bias=5.5)
    ocfg, net = _net(dev, sd)
    xs = _groups(11, shapes)
    logits, out_lens, steps = net.forward_groups([x.to(dev) for x in xs])
    lg = logits.cpu()
    assert lg.shape[0] == sum(b for b, _ in shapes)
    row = 0
    for x, n, st in zip(xs, out_lens, steps):
        ref, ref_steps = parseq_forward(sd, ocfg, x, return_steps=True)
        got = lg[row : row + x.shape[0], :n]
        assert st == ref_steps and got.shape == ref.shape
        assert torch.equal(got.argmax(-1), ref.argmax(-1))
        assert (got - ref).abs().max().item() < LOGIT_TOL
        one = net(x.to(dev)).cpu()
        assert net.last_ar_steps == st
        assert torch.equal(one.argmax(-1), got.argmax(-1))
        assert (one - got).abs().max().item() < 1e-4  # kernel shape choice follows M: last bits only
        row += x.shape[0]
    again, _, _ = net.forward_groups([x.to(dev) for x in xs])
    assert torch.equal(again.cpu(), lg), "grouped forward must be bit-identical on repeat"


def test_grouped_forward_with_a_repetition_stopped_row_in_a_short_group(dev):
    """A row stopped by the repetition detector keeps an ARG-MAX, not <eos>, at its last position.  When its mini-batch is
    the short one of a grouped forward, the shared greedy loop runs on for the other groups; the refinement must still
    see exactly the context the row's own forward builds (models/parseq.py:264-278: tgt_in from logits[:, :-1] of ITS
    loop) - positions beyond the group's own step count are masked.  Checked against the oracle and the single call."""
    from oracle.parseq import parseq_forward
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    z = np.load(os.path.join(GOLD, "parseq_ref_rep.npz"))
    sd = parseq_state_dict(**ast.literal_eval(str(z["ckpt"])))
    ocfg, net = _net(dev, sd)
    rows = [torch.from_numpy(z["x"][i : i + 1]) for i in range(z["x"].shape[0])]
    rows += [synthetic_line_batch(40 + i, 1, w) for i, w in enumerate((96, 240, 480, 800))]
    steps = []
    for r in rows:
        net(r.to(dev))
        steps.append(net.last_ar_steps)
    short, long_ = int(np.argmin(steps)), int(np.argmax(steps))
    assert steps[short] < steps[long_] < 101, steps  # the short group is cut by the detector well before the loop ends
    xs = [rows[short], rows[long_]]
    logits, out_lens, got_steps = net.forward_groups([x.to(dev) for x in xs])
    assert list(got_steps) == [steps[short], steps[long_]]
    lg = logits.cpu()
    for k, x in enumerate(xs):
        ref, ref_steps = parseq_forward(sd, ocfg, x, return_steps=True)
        got = lg[k : k + 1, : out_lens[k]]
        assert ref_steps == got_steps[k] and got.shape == ref.shape
        assert torch.equal(got.argmax(-1), ref.argmax(-1)) and (got - ref).abs().max().item() < LOGIT_TOL
        one = net(x.to(dev)).cpu()
        assert (one - got).abs().max().item() < 1e-4


def test_grouped_forward_open_beta_no_refine(dev):
    """Per-op decoder path (D = 512) with ragged encoder memory, refine_iters = 0: each group returns the AR logits of
    the steps ITS loop would have run, although the shared loop runs until the slowest group is done."""
    from oracle.parseq import parseq_forward
    from yomitoku_amd.utils.synth import parseq_state_dict

    kw = dict(patch=(8, 8), enc_dim=512, dec_dim=512, num_tokens=7312, enc_depth=3)
    sd = parseq_state_dict(77, eos_bias=6.0, **kw)
    ocfg, net = _net(dev, sd, "parseq", enc_depth=3, refine_iters=0)
    xs = _groups(21, [(4, 224), (2, 96), (3, 400)])
    logits, out_lens, steps = net.forward_groups([x.to(dev) for x in xs])
    lg = logits.cpu()
    row = 0
    for x, n, st in zip(xs, out_lens, steps):
        ref, ref_steps = parseq_forward(sd, ocfg, x, return_steps=True)
        got = lg[row : row + x.shape[0], :n]
        assert st == ref_steps and n == ref_steps and got.shape == ref.shape
        assert (got - ref).abs().max().item() < LOGIT_TOL
        row += x.shape[0]
    print("open-beta grouped steps", steps)


def test_grouped_forward_open_beta_refine(dev):
    """Per-op decoder path with refinement (flash cross-attention over ragged memory, Lq = 101)."""
    from oracle.parseq import parseq_forward
    from yomitoku_amd.utils.synth import parseq_state_dict

    kw = dict(patch=(8, 8), enc_dim=512, dec_dim=512, num_tokens=7312, enc_depth=2)
    sd = parseq_state_dict(79, eos_bias=6.0, **kw)
    ocfg, net = _net(dev, sd, "parseq", enc_depth=2)
    xs = _groups(31, [(3, 160), (2, 640)])
    logits, out_lens, steps = net.forward_groups([x.to(dev) for x in xs])
    lg = logits.cpu()
    row = 0
    for x, n, st in zip(xs, out_lens, steps):
        ref, ref_steps = parseq_forward(sd, ocfg, x, return_steps=True)
        got = lg[row : row + x.shape[0], :n]
        assert st == ref_steps and got.shape == ref.shape
        assert torch.equal(got.argmax(-1), ref.argmax(-1))
        assert (got - ref).abs().max().item() < LOGIT_TOL
        row += x.shape[0]


def test_legacy_parseq_tiny_geometry_head_dim_46(dev):
    """The old `parseq-tiny` (cfg_text_recognizer_parseq_tiny.py: D = 368, 8 heads -> head dim 46, 8 x 16 patches, 400 px
    canvas, 50 characters): rows are only 8 B aligned, so attention takes the one-wave-per-query kernel with scalar key
    loads and the decoder the per-op path.  Single call and grouped (ragged) call vs the oracle."""
    from oracle.parseq import make_cfg, parseq_forward
    from yomitoku_amd.nets import PARSeq
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    geo = dict(patch=(8, 16), enc_dim=368, dec_dim=368, num_tokens=7121, max_label_length=50, img_size=(32, 400))
    sd = parseq_state_dict(1238, enc_depth=3, eos_bias=6.0, **geo)
    ocfg = make_cfg(patch=(8, 16), enc_dim=368, enc_heads=8, enc_depth=3, dec_dim=368, dec_heads=8, num_tokens=7121,
                    max_label_length=50, img_size=(32, 400))
    cfg = {"num_tokens": 7121, "max_label_length": 50, "refine_iters": 1, "decode_ar": 1, "repetition_stop": True,
           "data": {"img_size": [32, 400]},
           "encoder": {"patch_size": [8, 16], "num_heads": 8, "embed_dim": 368, "mlp_ratio": 4, "depth": 3},
           "decoder": {"embed_dim": 368, "num_heads": 8, "mlp_ratio": 4, "depth": 1}}
    net = PARSeq(cfg).load_state_dict(sd).to(dev)
    xs = [synthetic_line_batch(61, 3, 400), synthetic_line_batch(62, 2, 160)]
    for x in xs:
        ref, steps = parseq_forward(sd, ocfg, x, return_steps=True)
        out = net(x.to(dev)).cpu()
        assert net.last_ar_steps == steps and out.shape == ref.shape
        assert torch.equal(out.argmax(-1), ref.argmax(-1)) and (out - ref).abs().max().item() < LOGIT_TOL
    logits, out_lens, steps = net.forward_groups([x.to(dev) for x in xs])
    row = 0
    for x, n, st in zip(xs, out_lens, steps):
        ref, ref_steps = parseq_forward(sd, ocfg, x, return_steps=True)
        got = logits[row : row + x.shape[0], :n].cpu()
        assert st == ref_steps and torch.equal(got.argmax(-1), ref.argmax(-1)) and (got - ref).abs().max().item() < LOGIT_TOL
        row += x.shape[0]


@pytest.mark.parametrize("refine", [1, 0])
def test_fused_step_rows_per_block_agree_bit_for_bit(dev, refine):
    """The fused greedy step for 2 ... 4 samples per block (forwards with more rows than resident blocks) against the
    one-sample-per-block instance: same chain of operations per value - the matvec's K split over thread groups, the
    attention's key rows dealt to 16 virtual waves and merged in their order - so logits, tokens and step counts must be
    identical bits, including groups that finish early (frozen rows inside a live block), a row count that is no multiple
    of the rows per block, waves that serve no row (16 / 3 leaves one over), and refine_iters = 0 (the AR
    logits themselves are the output)."""
    from yomitoku_amd import _lib
    from yomitoku_amd.utils.synth import parseq_state_dict

    sd = parseq_state_dict(1235, eos_bias=5.5)
    _, net = _net(dev, sd, refine_iters=refine)
    xs = [x.to(dev) for x in _groups(23, [(7, 160), (3, 800), (9, 72), (1, 96), (6, 320), (5, 64)])]
    outs = {}
    try:
        for rows in (1, 2, 3, 4):
            _lib.debug_option("dec_rows", rows)
            logits, out_lens, steps = net.forward_groups(xs)
            outs[rows] = (logits.cpu(), list(out_lens), list(steps))
    finally:
        _lib.debug_option("dec_rows", 0)
    assert len(set(outs[1][2])) > 1, "the groups should stop at different steps (frozen rows next to live ones)"
    for rows in (2, 3, 4):
        assert outs[rows][1] == outs[1][1] and outs[rows][2] == outs[1][2]
        row = 0
        for x, n in zip(xs, outs[1][1]):  # positions past a group's out_len are not part of the result
            a, b = outs[rows][0][row : row + x.shape[0], :n], outs[1][0][row : row + x.shape[0], :n]
            assert torch.equal(a, b), f"{rows} rows per block differ from the one-row kernel"
            row += x.shape[0]
    # a single mini-batch (no group tables): rows per block on the plain entry point
    x = xs[0]
    try:
        _lib.debug_option("dec_rows", 1)
        one = net(x).cpu()
        _lib.debug_option("dec_rows", 3)
        four = net(x).cpu()
    finally:
        _lib.debug_option("dec_rows", 0)
    assert torch.equal(one, four)  # nets.PARSeq.forward returns logits[:, :out_len]


def test_row_max_head_gives_the_same_tokens_as_arg_max_over_stored_logits(dev):
    """With a refinement pass to follow, the greedy loop's vocabulary head keeps only (max, column) per 64-column tile
    (EPI_ROWMAX) instead of writing B x 7119 logits per step.  (value, lowest column) is a total order, so tokens, step
    counts, repetition cuts and therefore the refined logits must be exactly those of the stored-logits form - compared
    here on a grouped forward with early-finishing groups, and on a single batch."""
    from yomitoku_amd import _lib
    from yomitoku_amd.utils.synth import parseq_state_dict

    sd = parseq_state_dict(1235, eos_bias=5.5)
    _, net = _net(dev, sd)
    xs = [x.to(dev) for x in _groups(53, [(7, 160), (3, 800), (9, 72), (1, 96), (6, 320), (5, 64), (40, 128)])]
    outs = {}
    try:
        for flag in (1, 0):
            _lib.debug_option("parseq_no_rowmax", flag)
            logits, out_lens, steps = net.forward_groups(xs)
            single = net(xs[0])
            outs[flag] = (logits.cpu(), list(out_lens), list(steps), single.cpu(), net.last_ar_steps)
    finally:
        _lib.debug_option("parseq_no_rowmax", 0)
    a, b = outs[1], outs[0]
    assert a[1] == b[1] and a[2] == b[2] and a[4] == b[4]
    # the head GEMM of the row-max form always takes 64 x 64 tiles (never split-K), so AR logits may differ from the stored
    # form's in their last bits - tokens may not; the refined logits then agree to rounding
    assert torch.equal(a[0].argmax(-1), b[0].argmax(-1)) and (a[0] - b[0]).abs().max().item() < 1e-4
    assert torch.equal(a[3].argmax(-1), b[3].argmax(-1)) and (a[3] - b[3]).abs().max().item() < 1e-4


def test_gemm_row_chunks_leave_the_forward_unchanged(dev):
    """a GEMM whose A view would pass 4 GiB (the fc2 input of a 2048-line forward) is cut into row chunks
    inside the library.  Forced here at 1024 rows per chunk on a forward of 3840 token rows and 2424 decoder rows: rows are
    independent, so tokens and step counts are those of the unchunked forward; logits differ only by what a smaller launch's
    tile / split-K choice reorders in its K sums."""
    from yomitoku_amd import _lib
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    sd = parseq_state_dict(1235, eos_bias=5.5)
    _, net = _net(dev, sd)
    x = synthetic_line_batch(11, 24, 160).to(dev)
    whole = net(x).cpu()
    steps = net.last_ar_steps
    _lib.debug_option("gemm_row_limit", 1024)
    try:
        parts = net(x).cpu()
        assert net.last_ar_steps == steps
    finally:
        _lib.debug_option("gemm_row_limit", 0)
    assert torch.equal(parts.argmax(-1), whole.argmax(-1))
    assert (parts - whole).abs().max().item() < 1e-4
    assert torch.equal(net(x).cpu(), whole)  # and the option is off again
